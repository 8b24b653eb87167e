#!/usr/bin/env python3
"""bench.py -- the compute_paths hot path on N MI355X GPUs (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the whole launch set of the workload: state init
from the (HBM-resident) launch directions, LoS pass, and num_bounces+1 launches of the
trace / shade / compaction kernels.  Inputs (scene, endpoints, launch directions) are
resident in HBM before the timed region; outputs (compact path records) stay in HBM, sharded
over the ranks exactly as at N = 1 -- rays are independent, the path itself has no exchange
step.  The collection of every rank's records on rank 0 (RCCL gather over xGMI,
hermespy_rt_amd.sharding) is measured right after the timed region and reported beside the
metric ("gather": ms, bytes, GB/s, and the throughput if it were serialised into every step);
`--gather-in-step` puts it inside the timed step instead.  Why it is not the default: one
MI355X produces ~385 GB/s of path records on this workload, the root of a gather can ingest
at most 7 x 153 GB/s over xGMI, so a gather-to-one-GPU of everything is bandwidth-bound at
~2.8 producer GPUs whatever the kernels do.

Workload at N = 1 is BASELINE.json configs[2] (the config its metric and target are quoted
on): simple_street_canyon_with_cars.hrt, 1 TX + 4 RX, 4M rays, 4 bounces, 3.5 GHz, endpoints
as fixed in SURVEY.md 8(d).  Scaling is WEAK: N GPUs trace an N-times denser Fibonacci
sphere (N x 4M rays), ray-sharded round-robin in 4096-path granules.

Prints ONE JSON line on rank 0.  metric = resolved propagation paths per second (scatter
records written, blocked ones included, + LoS entries), whole job; ray-triangle tests/s of
the brute-force algorithm is reported beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E peak


def workloads():
    from tests import configs as K
    return {"c1": K.C1, "c2": K.C2, "c3": K.C3, "c4": K.C4, "c5": K.C5,
            "c3_doppler": K.C3_DOPPLER}


def describe(c):
    return "%s, %d TX + %d RX, %d rays/TX, %d bounces, %.1f GHz" % (
        os.path.basename(c["scene_path"]), len(c["tx_pos"]), len(c["rx_pos"]), c["num_paths"],
        c["num_bounces"], c["f_ghz"])


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
    single-threaded) on a bounded sample of the workload, else our CPU port of it."""
    import ctypes
    from oracle import oracle
    out = {}
    ref_so = os.path.join(REPO, "oracle", "_ref", "libhrt_ref.so")
    # the single-threaded reference does 1.2e8 (this container) to 2.9e8 (GPU box host) tests/s
    # on C3; size the sample for about budget_s at 2e8: a sparser Fibonacci sphere of the same
    # scene/endpoints (10-15 s on the GPU box, well under a minute anywhere)
    T = len(oracle.flatten(oracle.read_hrt(c["scene_path"]))["tri_vtx"])
    per_ray = T * (1 + len(c["rx_pos"])) * c["num_bounces"] * len(c["tx_pos"]) * 0.5
    n_sample = int(min(c["num_paths"], max(10000, budget_s * 2.0e8 / max(per_ray, 1.0))))
    sc = dict(c, num_paths=n_sample)
    from tests import configs as K
    if os.path.exists(ref_so):
        from hermespy_rt_amd import abi
        lib = abi.bind_reference_abi(ctypes.CDLL(ref_so))
        t0 = time.time()
        r = abi.run_compute_paths(lib, *K.args(sc))
        dt = time.time() - t0
        recs = int(abi.written(r["scat"]["a_te_re"]).sum()) + len(c["rx_pos"]) * len(c["tx_pos"])
        out = dict(value=recs / dt, unit="paths/s", cores=1, kind="reference",
                   sample="%s (num_paths %d of %d), reference src/compute_paths.c built "
                          "gcc -O3 -ffp-contract=off, %.1f s" % (describe(sc), n_sample, c["num_paths"], dt))
        del r
    # our port, all host cores (an upper bound for an embarrassingly parallel CPU version)
    # the box's CPU share for one GPU is 16 threads
    nthr = min(16, oracle.lib().hrt_oracle_max_threads(), len(os.sched_getaffinity(0)))
    t0 = time.time()
    o = oracle.compute_paths(*K.args(sc), num_threads=nthr)
    dt = time.time() - t0
    recs = int(o["extras"]["live"][1:].sum()) * len(c["rx_pos"]) + len(c["rx_pos"]) * len(c["tx_pos"])
    port = dict(value=recs / dt, unit="paths/s", cores=nthr, kind="port",
                tests_per_s=o["extras"]["tests"] / dt,
                sample="%s (num_paths %d of %d), oracle/hrt_oracle.c OpenMP, %.1f s" % (
                    describe(sc), n_sample, c["num_paths"], dt))
    if out:
        out["tests_per_s"] = o["extras"]["tests"] / max(1e-9, float(out["sample"].split(",")[-1].split()[0]))
        out["port_all_cores"] = port
        return out
    return port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--gather-in-step", action="store_true",
                    help="N > 1: run the RCCL gather of all records to rank 0 inside every timed step")
    ap.add_argument("--time-every", type=int, default=4,
                    help="record the per-kernel HIP events on every n-th timed step (default 4)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: do not measure the gather at all")
    ap.add_argument("--gather-timeout", type=float, default=120.0,
                    help="N > 1: give up on the gather measurement after this many seconds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch hrt_selftest_math_kernel over 32M floats (known traffic: "
                         "128 MiB read + 128 MiB written, 4 B/lane) to calibrate PMC byte counters")
    ap.add_argument("--dropin", action="store_true",
                    help="instead of the HBM-resident bench: time the host-array drop-in C ABI "
                         "(hrt_compute_paths_ex: launch dirs on the host, H2D, trace, D2H, dense "
                         "scatter) once and print its phase times (the PCIe-inclusive figure)")
    args = ap.parse_args()

    if args.dropin:
        import torch  # noqa: F401
        from hermespy_rt_amd import abi, lib
        from tests import configs as K
        c = workloads()[args.workload]
        st = lib.Stats()
        t0 = time.time()
        abi.run_compute_paths(lib.load(), *K.args(c), with_rays=False, stats=st)
        wall = time.time() - t0
        nb = c["num_bounces"]
        paths = int(st.records) + len(c["rx_pos"]) * len(c["tx_pos"])
        print(json.dumps(dict(
            mode="dropin (host arrays in/out, PCIe inclusive)", workload=describe(c),
            t_total_s=st.t_total_s, t_setup_s=st.t_setup_s, t_launch_dirs_host_s=st.t_launch_dirs_s,
            t_device_incl_h2d_s=st.t_device_s, t_readback_and_dense_scatter_s=st.t_readback_s,
            wall_incl_python_alloc_s=wall, paths_per_s=paths / st.t_total_s,
            tests_per_s=int(st.tests) / st.t_total_s, live=[int(st.live[i]) for i in range(nb + 1)],
            records=int(st.records), records_unblocked=int(st.records_unblocked))))
        return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (args.gpus, args.gpus))
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def xreduce(t, op):
        """all_reduce that also works over gloo (host staging) in rehearsals"""
        if world == 1:
            return t
        if rehearse:
            c = t.cpu()
            dist.all_reduce(c, op=op)
            return c.to(t.device)
        dist.all_reduce(t, op=op)
        return t

    from hermespy_rt_amd.device import Tracer
    from hermespy_rt_amd import sharding

    base = workloads()[args.workload]
    c = dict(base, num_paths=base["num_paths"] * world)   # weak scaling: denser sphere
    tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                c["num_paths"], c["num_bounces"], rank=rank, world=world)
    gather = None
    gather_err = None
    if world > 1 and not args.no_gather:
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

    for _ in range(args.warmup):
        step()
    # per-kernel HIP events (one hrt_timer = the events of one step), recorded on the launch stream
    # INSIDE the timed region and read after it -- the steps run back to back, asynchronously.
    # Every 4th timed step carries them (all of them with --time-every 1): 17 event records per
    # step cost ~4 % of a 1.7 ms step, and the timed region is the metric.
    every = max(1, int(args.time_every))
    timers = [tr.new_timer() if k % every == 0 else None for k in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(timers[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_all = xreduce(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
    dt = float(t_all.item())
    bounce_ms, los_ms, compact_ms, shade_ms = [], [], [], []
    for t in timers:
        if t is None:
            continue
        r = tr.read_timer(t)
        los_ms.append(r["los_ms"])
        bounce_ms.append(r["trace_ms"])
        shade_ms.append(r["shade_ms"])
        compact_ms.append(sum(r["scan_ms"]))

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
    tm = np.asarray(bounce_ms, dtype=np.float64)          # [steps, nb+1] trace kernel
    sm = np.asarray(shade_ms, dtype=np.float64)           # [steps, nb+1] shade kernel
    bm = tm + sm
    kern_ms_step = float(bm.sum(axis=1).mean())
    n_launch = bm.shape[1]
    unb_local = unblocked
    B_local = algorithmic_bytes(w["live"], nrx, tr.num_local, unb_local, w["records"] - unb_local)
    ach = B_local / (kern_ms_step * 1e-3) / 1e9
    tests_local = w["tests"]
    # HBM bytes per launch from PMC counters: collected by profiles/collect_pmc.sh (separate
    # rocprofv3 passes) for exactly this workload at N = 1, committed in profiles/
    traffic, traffic_src = None, None
    try:
        pj = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        if world == 1 and args.workload in pj:
            traffic = pj[args.workload]["hbm_bytes_per_launch_avg"]
            traffic_src = pj[args.workload]["source"]
    except (OSError, ValueError, KeyError):
        pass
    # VALU issue utilisation of the two kernels (SURVEY 8d asks for the VALU fraction next to the
    # HBM one): from profiles/collect_valu.sh (own rocprofv3 --pmc pass), same workload, N = 1
    valu = None
    try:
        vj = json.load(open(os.path.join(REPO, "profiles", "pmc_valu.json")))
        if world == 1 and args.workload in vj:
            valu = dict(vj[args.workload]["kernels"], source=vj[args.workload]["source"])
    except (OSError, ValueError, KeyError):
        pass
    roofline = dict(bound="hbm", kernel="hrt_trace_kernel + hrt_shade_kernel (one bounce launch = the pair)",
                    achieved=ach, peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=B_local / n_launch,
                    avg_launch_ms=kern_ms_step / n_launch, launches_per_step=n_launch,
                    steps_with_kernel_events=int(bm.shape[0]),
                    per_launch_ms=[float(x) for x in bm.mean(axis=0)],
                    trace_kernel_ms=[float(x) for x in tm.mean(axis=0)],
                    shade_kernel_ms=[float(x) for x in sm.mean(axis=0)],
                    kernel_tests_per_s=tests_local / (kern_ms_step * 1e-3),
                    compaction_ms_per_step=float(np.mean(compact_ms)), los_ms=float(np.mean(los_ms)),
                    trace_variant=os.environ.get("HRT_TRACE_VARIANT", "default(2: packet culling)"),
                    valu=valu,
                    note="VALU-bound intersection work, not HBM-bound: see DESIGN.md section 6")

    kstats = None
    try:
        import ctypes
        from hermespy_rt_amd import lib as _l2
        arr = (ctypes.c_uint64 * 48)()
        if _l2.load().hrt_debug_kernel_stats(local_rank, arr, 0) == 0 and any(arr):
            kstats = [[int(arr[k * 16 + j]) for j in range(8)] for k in range(3)]
    except Exception:
        pass
    if rank == 0:
        out = dict(
            metric="resolved propagation paths/sec", value=paths * args.steps / dt,
            unit="paths/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=dt / args.steps * 1e3, higher_is_better=True, scaling="weak",
            vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload=describe(c), name=args.workload,
                        parallelism="ray-sharded x%d, round-robin 4096-path granules%s" % (
                            world, ", RCCL gather to rank 0 inside the step" if in_step else ""),
                        rays_total=c["num_paths"] * ntx),
            ray_tri_tests_per_sec=tests * args.steps / dt,
            nonzero_paths_per_sec=(unblk + nrx * ntx) * args.steps / dt,
            work=dict(live=live, records=records, records_unblocked=unblk, tests=tests),
            roofline=roofline)
        if kstats:
            out["kernel_stats_all_steps"] = dict(columns=["wave_traces", "usable_packets", "candidates", "stage2", "stage3", "exact", "heavy_packets", "heavy_candidates"], primary0=kstats[0], primary=kstats[1], shadow=kstats[2])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(base, args.cpu_budget_s)
    else:
        out = None

    # ---- the collection step, measured on its own (N > 1), LAST and under a watchdog: the
    # metric above is complete before the first collective of the gather is issued, and if the
    # gather hangs (it cannot be rehearsed over real xGMI links on a one-GPU box) every rank
    # leaves after `--gather-timeout` seconds and rank 0 still prints the line ----
    gather_info = None
    if gather is not None and not in_step:
        import threading

        def bail():
            if rank == 0:
                out["gather_error"] = "timeout after %.0f s" % args.gather_timeout
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(args.gather_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
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
            words = sum(sharding.export_words(gather.counts_all[r], tr.nb, tr.nrx)
                        for r in range(world) if r != 0) if rank == 0 else 0
            gather_info = dict(ms=float(g_dt.item()) * 1e3, bytes_into_root=int(words) * 4)
        except Exception as e:
            gather_err = "run: %r" % (e,)
        dog.cancel()
    if rank == 0:
        if gather_info:
            gms = gather_info["ms"]
            gather_info["GBps_into_root"] = gather_info["bytes_into_root"] / max(gms, 1e-9) / 1e6
            gather_info["value_if_serialised_into_step"] = paths / (dt / args.steps + gms * 1e-3)
            gather_info["what"] = ("all ranks' packed path records -> rank 0, RCCL grouped send/recv over "
                                   "xGMI, measured after the timed region (3 runs, max over ranks)")
            out["gather"] = gather_info
        if gather_err:
            out["gather_error"] = gather_err
        print(json.dumps(out), flush=True)
    if world > 1:
        dog2 = None
        try:
            import threading
            dog2 = threading.Timer(60.0, lambda: os._exit(0))   # the line is out: never hang on teardown
            dog2.daemon = True
            dog2.start()
            dist.barrier()
            dist.destroy_process_group()
        finally:
            if dog2 is not None:
                dog2.cancel()


if __name__ == "__main__":
    main()
