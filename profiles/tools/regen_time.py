import sys, time, ctypes as C
sys.path.insert(0, ".")
import torch
import hermespy_rt_amd
from hermespy_rt_amd.device import Tracer
from hermespy_rt_amd import lib as _lib
from hermespy_rt_amd.workloads import WORKLOADS
c = WORKLOADS["c3"]
tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"], c["num_paths"], c["num_bounces"])
L = tr.L
stream = C.c_void_p(torch.cuda.current_stream(tr.device).cuda_stream)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
d = lambda: _lib.check(L.hrt_launch_dirs_device(C.byref(tr.shard), C.c_void_p(tr.dirs.data_ptr()), tr.device.index, stream, None), "d")
o = lambda: _lib.check(L.hrt_launch_order_device(C.byref(tr.shard), C.c_void_p(tr.order.data_ptr()), tr.device.index, stream), "o")
def p():
    tr.dirs_launch = tr.dirs[tr.order.to(torch.int64) & 0xFFFFFFFF].contiguous()
print("dirs %.3f ms  order %.3f ms  permute %.3f ms  all %.3f ms  trace %.3f" % (t(d), t(o), t(p), t(tr.regen_launch_tables), t(tr.trace)))
