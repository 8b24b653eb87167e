#!/usr/bin/env python3
"""hrt_compute_paths_list on a named workload, timed by the library (GPU box):
    python profiles/list_wall.py [c3] [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hermespy_rt_amd import abi, lib, workloads as W   # noqa: E402

c = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
L = lib.load()
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    st = lib.Stats()
    pl = abi.run_compute_paths_list(L, *W.args(c), stats=st)
    print("list call %d: total %.4f s  device %.4f  readback+fill %.4f  (%d records)"
          % (k, st.t_total_s, st.t_device_s, st.t_readback_s, pl["rx"].size), flush=True)
    del pl
