"""Warm drop-in calls (hrt_compute_paths_ex) of one workload, phase by phase, output arrays allocated ONCE
(np.empty, touched by the first call): python profiles/dropin_calls.py <workload> [calls=3]
HRT_WORKSPACE_BYTES / HRT_POOL_MAX_BYTES are honoured (number of batches in the last column); a third argument
"rays" requests the RaysInfo snapshots as the reference's own callers do."""
import sys
sys.path.insert(0, ".")
from hermespy_rt_amd import abi, lib, workloads as W
L = lib.load()
name = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rays = len(sys.argv) > 3 and sys.argv[3] == "rays"
c = W.WORKLOADS[name]
for k in range(calls):
    st = lib.Stats()
    abi.run_compute_paths(L, *W.args(c), with_rays=rays, stats=st)
    print(name, "rays" if rays else "", "call %d: tot %.1f setup %.1f tables %.1f dev %.1f rb %.1f ms, %d batches" %
          (k, 1e3 * st.t_total_s, 1e3 * st.t_setup_s, 1e3 * st.t_launch_dirs_s, 1e3 * st.t_device_s, 1e3 * st.t_readback_s, st.num_batches), flush=True)
