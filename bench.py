#!/usr/bin/env python3
"""bench.py -- the compute_paths hot path on N MI355X GPUs (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3]

With --gpus N > 1 and no torch.distributed environment, bench.py launches its own N rank
processes (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1,
started BEFORE this process touches the GPU) and passes their output through; under
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE set) it is a rank.

A "step" is one pass of the hot path over the whole launch set of the workload: state init from
the (HBM-resident) launch directions, LoS pass, and num_bounces+1 launches of the trace / shade
kernels.  SCOPE of `value`: inputs (scene, endpoints, launch directions in launch order) are
resident in HBM before the timed region and outputs (compact path records) stay in HBM, sharded
over the ranks -- it is the steady-state kernel rate.  What a caller of the drop-in pays on top is
reported beside it, measured in the same process right after the timed region (N = 1):

    step_incl_launch_ms   the step with the launch directions and launch order generated inside it
                          (they are part of compute_paths in the reference, src/compute_paths.c:442-456)
    end_to_end            hrt_compute_paths_ex on host arrays, cold and warm, with its phase split
                          (setup, launch tables, device, D2H + dense scatter) -- the PCIe-inclusive figure
    sustained             the same step repeated for >= 10 s of GPU time (the 20-step timed region is
                          35 ms; this is the region a utilisation sampler can see)

N > 1: ray-sharded round-robin in 4096-path granules.  --scaling weak (default): N GPUs trace an
N-times denser Fibonacci sphere (the workload's ray count per GPU); --scaling strong: the
workload's ray count is the total -- `--workload c4 --gpus 4 --scaling strong` is BASELINE
configs[3], `--workload c5 --gpus 8 --scaling strong` configs[4]; at N = 1 both are the same run.
`value` has no exchange step in it (rays are independent); the collection of the records is
measured right after and reported as first-class numbers:

    value_with_gather     every rank's packed records -> rank 0 over RCCL (xGMI), serialised into the step
    value_with_d2h        every rank copies its own packed records to its host over its own PCIe link
                          (--collect d2h / both; no root)

Prints ONE JSON line on rank 0.  metric = resolved propagation paths per second (scatter records
written, blocked ones included, + LoS entries), whole job; ray-triangle tests/s of the brute-force
algorithm and the non-zero (unblocked) paths/s are reported beside it.  A hang in a collective
ends the run with a NON-ZERO exit code after the line (with `gather_error`) has been printed.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E peak

_print_lock = threading.Lock()
_printed = False


def emit(out):
    """rank 0's one JSON line, printed exactly once (the watchdogs race the main thread)"""
    global _printed
    with _print_lock:
        if _printed:
            return
        _printed = True
        print(json.dumps(out), flush=True)


def algorithmic_bytes(live, nrx, n_launch_rays, rec_unblocked, rec_blocked):
    """B = sum_b (44 A_b + 44 H_b) + 12 np + 36 R_unblocked + 20 R_blocked.
    44 = Ray 24 + gains 16 + tau 4; a record is 36 B (4 gains, tau, dir_rx, freq_shift),
    a blocked record 20 B."""
    nb = len(live) - 1
    b = 0
    for k in range(nb):
        b += 44 * live[k] + 44 * live[k + 1]
    return b + 12 * n_launch_rays + 36 * rec_unblocked + 20 * rec_blocked


def cpu_baseline(c, budget_s=20.0):
    """The reference itself (oracle/_ref, built in place from the reference sources; it is
    single-threaded) on a bounded sample of the workload, and our CPU port of it on all cores."""
    from oracle import oracle                      # checker: only this leg of bench.py uses it
    from tests import refabi
    from hermespy_rt_amd import abi, workloads as W
    out = {}
    # the single-threaded reference does 1.2e8 (build container) to 2.9e8 (GPU box host) tests/s
    # on C3; size the sample for about budget_s at 2e8: a sparser Fibonacci sphere of the same
    # scene/endpoints (10-15 s on the GPU box, well under a minute anywhere)
    T = len(oracle.flatten(oracle.read_hrt(c["scene_path"]))["tri_vtx"])
    per_ray = T * (1 + len(c["rx_pos"])) * c["num_bounces"] * len(c["tx_pos"]) * 0.5
    n_sample = int(min(c["num_paths"], max(10000, budget_s * 2.0e8 / max(per_ray, 1.0))))
    sc = dict(c, num_paths=n_sample)
    ref_dt = None
    if refabi.available():
        lib = refabi.load()
        t0 = time.time()
        r = abi.run_compute_paths(lib, *W.args(sc))
        ref_dt = time.time() - t0
        recs = int(abi.written(r["scat"]["a_te_re"]).sum()) + len(c["rx_pos"]) * len(c["tx_pos"])
        out = dict(value=recs / ref_dt, unit="paths/s", cores=1, kind="reference", seconds=ref_dt,
                   sample="%s (num_paths %d of %d), reference src/compute_paths.c built "
                          "gcc -O3 -ffp-contract=off, %.1f s" % (W.describe(sc), n_sample, c["num_paths"], ref_dt))
        del r
    # our port, all host cores (an upper bound for an embarrassingly parallel CPU version);
    # the box's CPU share for one GPU is 16 threads
    nthr = min(16, oracle.lib().hrt_oracle_max_threads(), len(os.sched_getaffinity(0)))
    t0 = time.time()
    o = oracle.compute_paths(*W.args(sc), num_threads=nthr)
    dt = time.time() - t0
    recs = int(o["extras"]["live"][1:].sum()) * len(c["rx_pos"]) + len(c["rx_pos"]) * len(c["tx_pos"])
    port = dict(value=recs / dt, unit="paths/s", cores=nthr, kind="port", seconds=dt,
                tests_per_s=o["extras"]["tests"] / dt,
                sample="%s (num_paths %d of %d), oracle/hrt_oracle.c OpenMP, %.1f s" % (
                    W.describe(sc), n_sample, c["num_paths"], dt))
    if out:
        out["tests_per_s"] = o["extras"]["tests"] / max(1e-9, ref_dt)
        out["port_all_cores"] = port
        return out
    return port


def self_launch(args):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks ourselves.  This
    process has not touched the GPU (no HIP call, no torch.cuda query), so starting children is safe;
    it never replaces itself with another program."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dropin_once(c, lib_, abi, W, with_rays=False):
    st = lib_.Stats()
    t0 = time.time()
    abi.run_compute_paths(lib_.load(), *W.args(c), with_rays=with_rays, stats=st)
    wall = time.time() - t0
    paths = int(st.records) + len(c["rx_pos"]) * len(c["tx_pos"])
    host_bytes = 4 * 9 * len(c["rx_pos"]) * len(c["tx_pos"]) * c["num_bounces"] * c["num_paths"]
    return dict(t_total_s=st.t_total_s, t_setup_s=st.t_setup_s, t_launch_tables_s=st.t_launch_dirs_s,
                t_device_s=st.t_device_s, t_readback_and_dense_scatter_s=st.t_readback_s,
                wall_incl_python_alloc_s=wall, paths_per_s=paths / st.t_total_s,
                tests_per_s=int(st.tests) / st.t_total_s, records=int(st.records),
                records_unblocked=int(st.records_unblocked), dense_host_bytes=host_bytes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--gather-in-step", action="store_true",
                    help="N > 1: run the RCCL gather of all records to rank 0 inside every timed step")
    ap.add_argument("--collect", choices=("gather", "d2h", "both", "none"), default="both",
                    help="N > 1: which collection step(s) to measure after the timed region")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = every GPU traces the workload's ray count (an N-times denser sphere in "
                         "all); strong = the workload's ray count is the TOTAL, ray-sharded over the N GPUs -- "
                         "`--workload c4 --gpus 4 --scaling strong` is BASELINE configs[3] (16 M rays in all), "
                         "`--workload c5 --gpus 8 --scaling strong` configs[4] (64 M)")
    ap.add_argument("--event-steps", type=int, default=10,
                    help="steps of the separate, untimed pass that carries the per-kernel HIP events")
    ap.add_argument("--no-gather", action="store_true", help="same as --collect none")
    ap.add_argument("--gather-timeout", type=float, default=120.0,
                    help="N > 1: give up on a collection measurement after this many seconds (exit code 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="N = 1: skip the drop-in / launch-inclusive / sustained measurements")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--sustain-s", type=float, default=10.0,
                    help="GPU seconds of the sustained region (long enough for a utilisation sampler)")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch hrt_selftest_math_kernel over 32M floats (known traffic: "
                         "128 MiB read + 128 MiB written, 4 B/lane) to calibrate PMC byte counters")
    ap.add_argument("--dropin", action="store_true",
                    help="only time the host-array drop-in C ABI (hrt_compute_paths_ex), cold and warm")
    args = ap.parse_args()
    if args.no_gather:
        args.collect = "none"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from hermespy_rt_amd import workloads as W
    if args.dropin:
        import torch  # noqa: F401
        from hermespy_rt_amd import abi, lib
        c = W.WORKLOADS[args.workload]
        cold = dropin_once(c, lib, abi, W)
        warm = dropin_once(c, lib, abi, W)
        print(json.dumps(dict(mode="dropin (host arrays in/out, PCIe inclusive)", workload=W.describe(c),
                              cold=cold, warm=warm)))
        return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback)")
    # HRT_BENCH_REHEARSE=1: N ranks share GPU 0 and talk over gloo -- a functional rehearsal of
    # the N > 1 code path on a one-GPU box (timings are meaningless there)
    rehearse = os.environ.get("HRT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if rehearse else "nccl"
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def xreduce(t, op):
        """all_reduce that also works over gloo (host staging) in rehearsals"""
        if world == 1:
            return t
        if rehearse:
            c_ = t.cpu()
            dist.all_reduce(c_, op=op)
            return c_.to(t.device)
        dist.all_reduce(t, op=op)
        return t

    from hermespy_rt_amd.device import Tracer
    from hermespy_rt_amd import sharding

    base = W.WORKLOADS[args.workload]
    # weak scaling: an N-times denser sphere, the workload's ray count per GPU; strong scaling: the
    # workload's ray count in all, sharded (BASELINE configs[3] / [4] are stated this way)
    c = dict(base, num_paths=base["num_paths"] * (world if args.scaling == "weak" else 1))
    tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                c["num_paths"], c["num_bounces"], rank=rank, world=world)
    gather = None
    gather_err = None
    if world > 1 and args.collect in ("gather", "both") or (world > 1 and args.gather_in_step):
        try:
            gather = sharding.RecordGather(tr)
        except Exception as e:   # never let the collection step take the metric down
            gather_err = "init: %r" % (e,)
    in_step = gather is not None and args.gather_in_step

    def step(timer=None):
        if timer is None:
            tr.trace()
        else:
            tr.trace_with_timer(timer)   # HIP events around every kernel, no host sync
        if in_step:
            gather.run()

    if args.calibrate:
        import ctypes
        from hermespy_rt_amd import lib as _l
        n = 32 << 20
        x = np.linspace(0.0, 1.0, n, dtype=np.float32)
        y = np.empty_like(x)
        f32p = ctypes.POINTER(ctypes.c_float)
        _l.check(_l.load().hrt_selftest_math(local_rank, 1, x.ctypes.data_as(f32p), y.ctypes.data_as(f32p), n))

    # A VOID step (a fused launch / the chain kernel gave up waiting: the GPU is shared with other such kernels; the
    # library then switches that kernel off and callers trace again) must not be timed as if it were work: the error
    # word is looked at behind the region, on every rank, and the region is timed again on the smaller kernels.
    void_retimed = 0
    from hermespy_rt_amd import lib as _libm
    for attempt in range(3):
        fb0 = int(_libm.load().hrt_fallback_state())
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t_all = xreduce(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
        dt = float(t_all.item())
        # (a void step anywhere in the region: the last step's error word, or a kernel switched off meanwhile)
        void = 1.0 if ((tr.error_word() & 0x300) or int(_libm.load().hrt_fallback_state()) != fb0) else 0.0
        void = float(xreduce(torch.tensor([void], dtype=torch.float64, device=dev), dist.ReduceOp.MAX).item())
        if not void:
            break
        void_retimed += 1
    # per-kernel HIP events (one hrt_timer = the events of one step, recorded on the launch stream,
    # read afterwards): a SEPARATE untimed pass right after the timed region, every step of it
    # instrumented -- the timed steps carry no events (they cost ~4 % of a 1.6 ms step)
    timers = [tr.new_timer() for _ in range(max(1, args.event_steps))]
    for t in timers:
        step(t)
    torch.cuda.synchronize()
    bounce_ms, los_ms, compact_ms, shade_ms, records_ms = [], [], [], [], []
    for t in timers:
        r = tr.read_timer(t)
        los_ms.append(r["los_ms"])
        bounce_ms.append(r["trace_ms"])
        shade_ms.append(r["shade_ms"])
        compact_ms.append(sum(r["scan_ms"]))
        records_ms.append(r["records_ms"])

    # ---- work done (identical every step) ----
    counts = tr.counts()
    w = tr.work(counts)
    nrx, ntx, nb = tr.nrx, tr.ntx, tr.nb
    unblocked = 0
    for b in range(nb):
        n = int(counts[b + 1])
        if n:
            unblocked += int(tr.records(b, n)["unblocked"].sum().item())
    local = torch.tensor([w["records"], w["tests"] - nrx * ntx * tr.num_tri, unblocked] + w["live"],
                         dtype=torch.float64, device=dev)
    local = xreduce(local, dist.ReduceOp.SUM)
    tot = [int(x) for x in local.tolist()]
    records, tests, unblk, live = tot[0], tot[1] + nrx * ntx * tr.num_tri, tot[2], tot[3:]
    paths = records + nrx * ntx

    # ---- roofline of the dominant kernels, rank 0's launches.  One launch of "the bounce" is
    # the pair hrt_trace_kernel (intersection) + hrt_shade_kernel (records, Fresnel, reflect):
    # the algorithmic bytes of SURVEY 8(d) are those of the pair, so is the duration. ----
    tm = np.asarray(bounce_ms, dtype=np.float64)          # [steps, nb+1] trace kernel (patch tables: primary rays only)
    sm = np.asarray(shade_ms, dtype=np.float64)           # [steps, nb+1] shade kernel
    rm = np.asarray(records_ms, dtype=np.float64)         # [steps, nb+1] records kernel (patch tables; else 0)
    bm = tm + sm + rm
    kern_ms_step = float(bm.sum(axis=1).mean())
    n_launch = bm.shape[1]
    unb_local = unblocked
    B_local = algorithmic_bytes(w["live"], nrx, tr.num_local, unb_local, w["records"] - unb_local)
    ach = B_local / (kern_ms_step * 1e-3) / 1e9
    tests_local = w["tests"]
    # HBM bytes per launch from PMC counters: collected by profiles/collect_pmc.sh (separate
    # rocprofv3 passes) for exactly this workload at N = 1, committed in profiles/.  The JSON carries
    # the hash of the kernel source it was measured on: `traffic_stale` says whether that is still
    # the source of this run.
    kern_sha = hashlib.sha256(b"".join(open(os.path.join(REPO, "hermespy-rt_amd", "csrc", f), "rb").read() for f in ("hrt_kernels.hip", "hrt_fused_body.inc"))).hexdigest()[:16]
    traffic, traffic_src, traffic_stale = None, None, None
    try:
        pj = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        if world == 1 and args.workload in pj:
            traffic = pj[args.workload]["hbm_bytes_per_launch_avg"]
            traffic_src = pj[args.workload]["source"]
            traffic_stale = pj[args.workload].get("kernels_sha16") != kern_sha
    except (OSError, ValueError, KeyError):
        pass
    # VALU issue utilisation of the two kernels (SURVEY 8d asks for the VALU fraction next to the
    # HBM one): from profiles/collect_valu.sh (own rocprofv3 --pmc pass), same workload, N = 1
    valu = None
    try:
        vj = json.load(open(os.path.join(REPO, "profiles", "pmc_valu.json")))
        if world == 1 and args.workload in vj:
            valu = dict(vj[args.workload]["kernels"], source=vj[args.workload]["source"],
                        stale=vj[args.workload].get("kernels_sha16") != kern_sha)
    except (OSError, ValueError, KeyError):
        pass
    # which resource binds: the VALU issue rate of the dominant kernel (committed PMC pass) against
    # the HBM fraction of the path -- C3 is VALU-bound, the tiny tables are nearer HBM
    hbm_frac = ach / HBM_PEAK_GBS
    valu_frac = None
    if valu:
        fr = [(v["valu_insts_per_step"], v["issue_frac_vs_simd32_peak"]) for k, v in valu.items()
              if isinstance(v, dict) and "issue_frac_vs_simd32_peak" in v]
        if fr:
            valu_frac = max(fr)[1]   # of the kernel that issues the most instructions
    fused0 = os.environ.get("HRT_FUSE", "") not in ("0",)
    fused_all = fused0 and tr.num_tri <= 64
    patched = 64 < tr.num_tri <= 256 and "no_patch" not in os.environ.get("HRT_TUNE", "")
    kern_desc = ("hrt_fused_kernel (one kernel per launch: trace + shading + stable compaction)" if fused_all else
                 ("launch 0: hrt_fused_kernel; later launches: " if fused0 else "") +
                 ("hrt_records_kernel (shadow traces + scatter records; in the timed region on a second stream beside the "
                  "others) + hrt_image_kernel / hrt_trace_kernel (primary rays) + hrt_shade_kernel (Fresnel, reflection, "
                  "compaction): one bounce launch = the three; trace_kernel_ms = records + primary rays" if patched else
                  "hrt_trace_kernel + hrt_shade_kernel (one bounce launch = the pair)"))
    roofline = dict(bound=("valu" if (valu_frac is not None and valu_frac > hbm_frac) else "hbm"), kernel=kern_desc,
                    achieved=ach, peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=hbm_frac, hbm_frac=hbm_frac, valu_frac=valu_frac,
                    frac_of_step=B_local / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                    traffic=traffic, traffic_source=traffic_src,
                    traffic_stale=traffic_stale, traffic_measured_in_run=False, kernels_sha16=kern_sha,
                    algorithmic_bytes_per_launch=B_local / n_launch,
                    avg_launch_ms=kern_ms_step / n_launch, launches_per_step=n_launch,
                    steps_with_kernel_events=int(bm.shape[0]),
                    kernel_events="separate untimed pass after the timed region, every step instrumented",
                    per_launch_ms=[float(x) for x in bm.mean(axis=0)],
                    trace_kernel_ms=[float(x) for x in tm.mean(axis=0)],
                    shade_kernel_ms=[float(x) for x in sm.mean(axis=0)],
                    records_kernel_ms=[float(x) for x in rm.mean(axis=0)],
                    kernel_tests_per_s=tests_local / (kern_ms_step * 1e-3),
                    compaction_ms_per_step=float(np.mean(compact_ms)), los_ms=float(np.mean(los_ms)),
                    trace_variant=os.environ.get("HRT_TUNE", "auto (patch tables on 65-256 triangles; flat packet culling; trees on big sparse tables)"),
                    valu=valu,
                    note=("a fused launch reports its one kernel under trace_kernel_ms (shade_kernel_ms = 0); "
                          "frac is the north-star's HBM figure (algorithmic bytes / SUM of the kernels' own durations, "
                          "measured one after the other on one stream / 8 TB/s); frac_of_step the same bytes over the timed "
                          "step, in which the records kernels run beside the others; "
                          "valu_frac = VALU instructions issued x 2 cycles / SIMD-cycles of the busiest kernel"))

    if rm.sum() > 0:
        # the DOMINANT kernel on its own: hrt_records_kernel.  Algorithmic bytes of one launch b (the part
        # of the path this kernel performs): 44 B of state per entry in, 36 B per unblocked / 20 B per blocked record out
        rec_launches = [b for b in range(1, nb + 1) if w["live"][b] > 0]
        rec_bytes = 44.0 * sum(w["live"][b] for b in rec_launches) + 36.0 * unb_local + 20.0 * (w["records"] - unb_local)
        rec_ms = float(rm.sum(axis=1).mean())
        roofline["dominant_kernel"] = dict(
            name="hrt_records_kernel", launches_per_step=len(rec_launches), avg_launch_ms=rec_ms / max(1, len(rec_launches)),
            ms_per_step=rec_ms, algorithmic_bytes_per_step=rec_bytes,
            achieved=rec_bytes / (rec_ms * 1e-3) / 1e9, unit="GB/s", frac=rec_bytes / (rec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            what="HIP events around the kernel, one stream (the event pass); its VALU issue fraction is roofline.valu.records")

    kstats = None
    try:
        import ctypes
        from hermespy_rt_amd import lib as _l2
        arr = (ctypes.c_uint64 * 48)()
        if _l2.load().hrt_debug_kernel_stats(local_rank, arr, 0) == 0 and any(arr):
            kstats = [[int(arr[k * 16 + j]) for j in range(16)] for k in range(3)]
    except Exception:
        pass
    out = None
    if rank == 0:
        out = dict(
            metric="resolved propagation paths/sec", value=paths * args.steps / dt,
            unit="paths/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling=args.scaling if world > 1 else "weak",
            vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload=W.describe(c), name=args.workload,
                        parallelism="ray-sharded x%d (%s scaling), round-robin 4096-path granules%s" % (
                            world, args.scaling, ", RCCL gather to rank 0 inside the step" if in_step else ""),
                        rays_total=c["num_paths"] * ntx,
                        scope="steady-state kernels: scene, endpoints and launch directions resident in HBM, "
                              "compact records left in HBM; launch-table generation, H2D/D2H and the dense "
                              "scatter are NOT in `value` -- see step_incl_launch_ms and end_to_end"),
            ray_tri_tests_per_sec=tests * args.steps / dt,
            nonzero_paths_per_sec=(unblk + nrx * ntx) * args.steps / dt,
            work=dict(live=live, records=records, records_unblocked=unblk, tests=tests),
            roofline=roofline)
        # kernels this process switched off after a timeout (0: none; bit 0 fused launches, bit 1 the chain kernel:
        # the GPU was shared) and how often the timed region was timed again because a step in it was void
        out["fallback_state"] = int(_libm.load().hrt_fallback_state())
        out["void_regions_retimed"] = void_retimed
        if kstats:
            out["kernel_stats_all_steps"] = dict(columns=["wave_traces", "usable_packets", "candidates", "stage2", "stage3", "exact", "c6", "c7"], primary0=kstats[0], primary=kstats[1], shadow=kstats[2])

    # ---- N = 1: what the steady-state number leaves out, measured in this process ----
    if world == 1 and not args.no_end_to_end:
        try:
            # (a) sustained: the same step for >= sustain_s of GPU time
            n_sus = max(args.steps, int(args.sustain_s / max(dt / args.steps, 1e-6)) + 1)
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            for _ in range(n_sus):
                tr.trace()
            torch.cuda.synchronize()
            s_dt = time.perf_counter() - s0
            out["sustained"] = dict(steps=n_sus, seconds=s_dt, ms_per_step=s_dt / n_sus * 1e3,
                                    paths_per_s=paths * n_sus / s_dt)
            # (b) the step with the launch tables generated inside it (device generators)
            n_li = max(3, min(args.steps, 20))
            tr.regen_launch_tables()
            torch.cuda.synchronize()
            l0 = time.perf_counter()
            for _ in range(n_li):
                tr.regen_launch_tables()
                tr.trace()
            torch.cuda.synchronize()
            l_dt = (time.perf_counter() - l0) / n_li
            out["step_incl_launch_ms"] = l_dt * 1e3
            out["step_incl_launch"] = dict(ms=l_dt * 1e3, paths_per_s=paths / l_dt, steps=n_li,
                                           what="hrt_launch_dirs_device + hrt_launch_order_device + permutation "
                                                "of the direction table into launch order + the step")
            # (b') what ONE rank of a strong-scaling run does: the step of rank 0's round-robin shard of this
            # launch set at world 2 / 4 / 8, measured on this GPU (a prediction the driver's N-GPU run can be held
            # against: N x speed-up = world_1 / world_N if nothing but the shard's own step limits a rank)
            shard_ms = {}
            for w_ in (2, 4, 8):
                ts = Tracer(base["scene_path"], base["rx_pos"], base["tx_pos"], base["rx_vel"], base["tx_vel"],
                            base["f_ghz"], base["num_paths"], base["num_bounces"], rank=0, world=w_)
                for _ in range(3):
                    ts.trace()
                torch.cuda.synchronize()
                q0 = time.perf_counter()
                for _ in range(20):
                    ts.trace()
                torch.cuda.synchronize()
                shard_ms["world_%d" % w_] = (time.perf_counter() - q0) / 20 * 1e3
                ts.close()
                del ts
            out["strong_scaling_shard_ms"] = dict(world_1=dt / args.steps * 1e3, **shard_ms,
                                                  what="step of rank 0's shard of this workload's launch set (total rays "
                                                       "fixed), one GPU")
            # (c) the drop-in ABI end to end (host arrays, PCIe inclusive): first call and second call
            from hermespy_rt_amd import abi, lib as _lib
            tr.close()
            del tr
            torch.cuda.empty_cache()
            cold = dropin_once(base, _lib, abi, W)
            warm = dropin_once(base, _lib, abi, W)
            # ... and the literal call of inc/compute_paths.h:59-74 as the reference's own callers make it
            # (test/test.c:56-72, compute_paths_pybind11.cpp:149-170): RaysInfo snapshots requested too
            with_rays = dropin_once(base, _lib, abi, W, with_rays=True)
            with_rays = dropin_once(base, _lib, abi, W, with_rays=True)
            out["end_to_end"] = dict(
                what="hrt_compute_paths_ex (the drop-in C ABI behind compute_paths): host arrays in, the "
                     "reference's dense arrays out; cold = first call in this process after the bench "
                     "(HIP already initialised), warm = second call; with_raysinfo = the warm call with the "
                     "RaysInfo snapshots requested as every caller of the reference does (second such call); "
                     "python_module_s = wall of hermespy_rt.compute_paths() (the pybind11 drop-in: array "
                     "allocation, the call, complex amplitudes), second and third call",
                cold=cold, warm=warm, with_raysinfo=with_rays)
            # ... and the same call with the result as ONE list of records (hrt_compute_paths_list):
            # the library's own clock of the third call (the ctypes wrapper copies the list into numpy, which is
            # not in it); skipped where those copies would not fit comfortably (C5: 45 GB of list)
            if int(warm["records"]) * 65 <= (8 << 30):
                try:
                    pl_t = []
                    for _ in range(3):
                        st_ = _lib.Stats()
                        pl_ = abi.run_compute_paths_list(_lib.load(), *W.args(base), stats=st_)
                        pl_t.append(dict(t_total_s=st_.t_total_s, t_device_s=st_.t_device_s,
                                         t_readback_and_fill_s=st_.t_readback_s, records=int(pl_["rx"].size)))
                        del pl_
                    out["end_to_end"]["path_list"] = dict(
                        pl_t[-1], what="hrt_compute_paths_list: the non-zero records as one list (14 arrays, 65 B "
                                       "per record) instead of the dense arrays; third call")
                except Exception as e:
                    out["end_to_end"]["path_list_error"] = repr(e)
            try:   # the Python surface HermesPy imports
                import hermespy_rt_amd as _pkg
                if _pkg.LIB_DIR not in sys.path:
                    sys.path.insert(0, _pkg.LIB_DIR)
                import hermespy_rt as _mod
                import numpy as _np
                f32 = lambda a: _np.array(a, dtype=_np.float32)   # noqa: E731
                walls = []
                for _ in range(3):
                    p0 = time.perf_counter()
                    r_ = _mod.compute_paths(base["scene_path"], f32(base["rx_pos"]), f32(base["tx_pos"]),
                                            f32(base["rx_vel"]), f32(base["tx_vel"]), base["f_ghz"],
                                            len(base["rx_pos"]), len(base["tx_pos"]), base["num_paths"],
                                            base["num_bounces"])
                    walls.append(time.perf_counter() - p0)
                    del r_
                out["end_to_end"]["python_module_s"] = walls[1:]
            except Exception as e:   # the module is optional at build time
                out["end_to_end"]["python_module_error"] = repr(e)
        except Exception as e:
            out["end_to_end_error"] = repr(e)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(base, args.cpu_budget_s)

    # ---- N > 1: the collection step, measured on its own, LAST and under a watchdog: the metric
    # above is complete before the first collective is issued; if a collective hangs every rank
    # leaves after `--gather-timeout` seconds, rank 0 prints the line with `gather_error`, and the
    # exit code is 3 ----
    def watchdog(what):
        def bail():
            if rank == 0:
                out["gather_error"] = "%s: timeout after %.0f s" % (what, args.gather_timeout)
                emit(out)
            os._exit(3)
        t = threading.Timer(args.gather_timeout, bail)
        t.daemon = True
        t.start()
        return t

    if world > 1:
        if rank == 0:
            out["collect"] = dict(backend=backend, n_ranks_seen_by_backend=dist.get_world_size(),
                                  n_ranks_seen_by_rccl=(dist.get_world_size() if backend == "nccl" else 0))
        if gather is not None and not in_step:
            dog = watchdog("gather")
            try:
                if os.environ.get("HRT_BENCH_FORCE_HANG") == "1":   # test hook: a collective that never returns
                    time.sleep(args.gather_timeout + 60)
                gather.run()                      # warm-up: allocates the receive buffers
                torch.cuda.synchronize()
                dist.barrier()
                g0 = time.perf_counter()
                n_g = 3
                for _ in range(n_g):
                    gather.run()
                torch.cuda.synchronize()
                dist.barrier()
                g_dt = xreduce(torch.tensor([(time.perf_counter() - g0) / n_g], dtype=torch.float64,
                                            device=dev), dist.ReduceOp.MAX)
                if rank == 0:
                    words = sum(sharding.export_words(gather.counts_all[r], tr.nb, tr.nrx)
                                for r in range(world) if r != 0)
                    gms = float(g_dt.item()) * 1e3
                    out["collect"]["gather"] = dict(
                        ms=gms, bytes_into_root=int(words) * 4, GBps_into_root=int(words) * 4 / max(gms, 1e-9) / 1e6,
                        what="all ranks' packed path records -> rank 0, grouped send/recv, measured after the "
                             "timed region (3 runs, max over ranks)")
                    out["value_with_gather"] = paths / (dt / args.steps + gms * 1e-3)
            except Exception as e:
                gather_err = "run: %r" % (e,)
            dog.cancel()
        if args.collect in ("d2h", "both"):
            dog = watchdog("d2h")
            try:
                cnt = tr.counts()
                pk = sharding.pack_export(tr, cnt)
                host = torch.empty(pk.numel(), dtype=torch.int32, pin_memory=True)
                host.copy_(pk, non_blocking=True)
                torch.cuda.synchronize()
                dist.barrier()
                h0 = time.perf_counter()
                n_h = 3
                for _ in range(n_h):
                    pk = sharding.pack_export(tr, cnt, pk)
                    host.copy_(pk, non_blocking=True)
                torch.cuda.synchronize()
                dist.barrier()
                h_dt = xreduce(torch.tensor([(time.perf_counter() - h0) / n_h], dtype=torch.float64,
                                            device=dev), dist.ReduceOp.MAX)
                nbytes = xreduce(torch.tensor([float(pk.numel() * 4)], dtype=torch.float64, device=dev),
                                 dist.ReduceOp.SUM)
                if rank == 0:
                    hms = float(h_dt.item()) * 1e3
                    out["collect"]["d2h"] = dict(
                        ms=hms, bytes_all_ranks=int(nbytes.item()), GBps_all_ranks=float(nbytes.item()) / max(hms, 1e-9) / 1e6,
                        what="every rank packs its records (device) and copies them to page-locked host "
                             "memory over its own PCIe link; no root (3 runs, max over ranks)")
                    out["value_with_d2h"] = paths / (dt / args.steps + hms * 1e-3)
            except Exception as e:
                gather_err = (gather_err + "; " if gather_err else "") + "d2h: %r" % (e,)
            dog.cancel()
    if rank == 0:
        if gather_err:
            out["gather_error"] = gather_err
        emit(out)
    if world > 1:
        dog2 = threading.Timer(60.0, lambda: os._exit(4))   # the line is out; a hung teardown is still a failure
        dog2.daemon = True
        dog2.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        finally:
            dog2.cancel()


if __name__ == "__main__":
    main()
