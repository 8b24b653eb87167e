"""C5 (the 8-GPU configuration) on ONE GPU: whole, and as 8 logical shards; per-bounce live counts of
the shards must add up to the whole's.  Run from the repo root on a GPU box:
    python profiles/c5_one_gpu.py      (writes gpurun_out/c5_one_gpu.json; round 1: profiles/r01_c5_one_gpu.json)"""
import sys, time, json
import numpy as np
import torch
sys.path.insert(0, ".")
from tests import configs as K
from hermespy_rt_amd.device import Tracer
c = K.C5
out = {}
def run(rank, world):
    tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                c["num_paths"], c["num_bounces"], rank=rank, world=world)
    tr.trace(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        tr.trace()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    cnt = [int(x) for x in tr.counts()]
    w = tr.work(np.asarray(cnt))
    del tr
    torch.cuda.empty_cache()
    return ms, cnt, w
tot = None
for r in range(8):
    ms, cnt, w = run(r, 8)
    print("shard", r, round(ms, 2), "ms", cnt, flush=True)
    tot = cnt if tot is None else [a + b for a, b in zip(tot, cnt)]
    out["shard%d" % r] = dict(ms=ms, live=cnt, records=w["records"], tests=w["tests"])
print("sum  ", tot, flush=True)
free, total = torch.cuda.mem_get_info()
print("free GB", free / 2**30, "total", total / 2**30, flush=True)
ms, cnt, w = run(0, 1)
print("whole", round(ms, 2), "ms", cnt, "records", w["records"], "tests", w["tests"], flush=True)
out["whole"] = dict(ms=ms, live=cnt, records=w["records"], tests=w["tests"])
out["sum_equals_whole"] = tot[:len(cnt) - 1] == cnt[:len(cnt) - 1]
print("sum == whole:", out["sum_equals_whole"])
json.dump(out, open("gpurun_out/c5_one_gpu.json", "w"))
