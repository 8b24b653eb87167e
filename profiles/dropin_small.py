"""Warm drop-in calls (hrt_compute_paths_ex) of the small-table workloads, phase by phase; run with
HRT_TUNE=no_rxt=1 too to see what the per-cell candidate masks cost (setup) and save (device)."""
import sys
sys.path.insert(0, ".")
from hermespy_rt_amd import abi, lib, workloads as W
L = lib.load()
for name in sys.argv[1:] or ["c1", "c2", "c4"]:
    c = W.WORKLOADS[name]
    rows = []
    for k in range(6):
        st = lib.Stats()
        abi.run_compute_paths(L, *W.args(c), with_rays=False, stats=st)
        rows.append((st.t_total_s, st.t_setup_s, st.t_launch_dirs_s, st.t_device_s, st.t_readback_s))
    print(name, " | ".join("tot %.2f setup %.2f tables %.2f dev %.2f rb %.2f ms" % tuple(1e3 * x for x in r) for r in rows[2:]), flush=True)
