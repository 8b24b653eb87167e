import sys, time
sys.path.insert(0, ".")
from tests import configs as K
from hermespy_rt_amd import abi, lib
L = lib.load()
c = K.C3
for k in range(3):
    st = lib.Stats()
    t0 = time.time()
    abi.run_compute_paths(L, *K.args(c), with_rays=False, stats=st)
    wall = time.time() - t0
    print("call %d: total %.3f setup %.3f dirs %.3f dev %.3f readback %.3f other %.3f wall(py alloc incl) %.3f" % (
        k, st.t_total_s, st.t_setup_s, st.t_launch_dirs_s, st.t_device_s, st.t_readback_s,
        st.t_total_s - st.t_setup_s - st.t_launch_dirs_s - st.t_device_s - st.t_readback_s, wall), flush=True)

print("-- hrt_compute_paths_list (records only, no dense arrays) --", flush=True)
for k in range(3):
    st = lib.Stats()
    t0 = time.time()
    P = abi.run_compute_paths_list(L, *K.args(c), stats=st)
    wall = time.time() - t0
    print("call %d: total %.3f setup %.3f dirs %.3f dev %.3f readback %.3f wall(py copies incl) %.3f  records %d" % (
        k, st.t_total_s, st.t_setup_s, st.t_launch_dirs_s, st.t_device_s, st.t_readback_s, wall, P["rx"].size), flush=True)
