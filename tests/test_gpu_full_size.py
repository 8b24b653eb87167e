"""BASELINE.json's FULL sizes on the GPU (device-resident API), checked without any dense
array or CPU run: per-bounce hit counts and record counts against the reference's
(tests/golden/full_size.json, recorded from the real reference), and an order-independent
64-bit checksum of every output array computed on the GPU from the compact path records --
sum over written slots of mix64(slot << 32 | value_bits) -- against the same checksum of the
reference's dense arrays.  Plus size-independent properties: records are a pure function of
the ray (two runs agree bit for bit although the compaction order differs), delays grow
along a path, unit-norm arrival directions."""
import json
import os

import numpy as np
import pytest

from . import configs as K

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_size.json")))
CASES = {"C1": K.C1, "C2": K.C2, "C3": K.C3, "C4_1M": K.small(K.C4, 1000000),
         "C5_1M": K.small(K.C5, 1000000)}
if "C4" in GOLD:
    CASES["C4"] = K.C4   # BASELINE configs[3] at full size: 2 TX x 8 M rays x 6 bounces (reference: 12 GB dense)


def _lsr(x, n):
    return (x >> n) & ((1 << (64 - n)) - 1)


def _mix64(x):
    x = (x ^ _lsr(x, 30)) * -4658895280553007687      # 0xBF58476D1CE4E5B9 as int64
    x = (x ^ _lsr(x, 27)) * -7723592293110705685      # 0x94D049BB133111EB as int64
    return x ^ _lsr(x, 31)


def _cs(slots, val_i32):
    import torch
    key = (slots << 32) | (val_i32.to(torch.int64) & 0xFFFFFFFF)
    return int(_mix64(key).sum().item()) & ((1 << 64) - 1)


def _tracer(c, **kw):
    from hermespy_rt_amd.device import Tracer
    return Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"],
                  c["f_ghz"], c["num_paths"], c["num_bounces"], **kw)


@pytest.mark.parametrize("name", list(CASES))
def test_full_size_counts_and_checksums(name):
    import torch
    c, g = CASES[name], GOLD[name]
    tr = _tracer(c)
    tr.trace()
    counts = tr.counts()
    nrx, ntx, nb, npth = tr.nrx, tr.ntx, tr.nb, tr.num_paths
    assert [int(x) for x in counts[1:nb + 1]] == g["hits_per_bounce"]
    M = (1 << 64) - 1
    cs = {k: 0 for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau", "directions_rx")}
    written = 0
    unblocked = 0
    for b in range(nb):
        n = int(counts[b + 1])
        if not n:
            continue
        h = tr.hits(b, n)
        ray = h["ray"].to(torch.int64) & 0xFFFFFFFF
        tx, p = tr.global_path(ray)
        r = tr.records(b, n)
        for rx in range(nrx):
            slot = ((rx * ntx + tx) * nb + b) * npth + p
            written += n
            for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau"):
                cs[k] = (cs[k] + _cs(slot, r[k][rx].view(torch.int32))) & M
            ub = r["unblocked"][rx]
            unblocked += int(ub.sum().item())
            for ci, dk in enumerate(("dirx", "diry", "dirz")):
                cs["directions_rx"] = (cs["directions_rx"] + _cs(slot[ub] * 3 + ci, r[dk][rx][ub].view(torch.int32))) & M
    assert unblocked == g["records_unblocked"]
    for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau"):
        assert written == g["arrays"][k]["written"], k
    assert unblocked == g["arrays"]["directions_rx"]["written"]
    # geometry-only arrays: bit-exact by construction
    assert cs["tau"] == g["arrays"]["tau"]["checksum"]
    assert cs["directions_rx"] == g["arrays"]["directions_rx"]["checksum"]
    # amplitudes: bit-exact too (device libm == host libm, and the device's double acos of the
    # incidence angle equals glibc's on every float input: tests/exhaustive_incidence.py)
    amp_ok = all(cs[k] == g["arrays"][k]["checksum"] for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im"))
    assert amp_ok, "amplitude checksum differs from the reference at full size"
    # LoS block
    los = tr.los()
    assert np.array_equal(los[:, :, 1].astype(np.float32).ravel(), np.asarray(g["los"]["a_te_re"], np.float32))
    assert np.array_equal(los[:, :, 2].astype(np.float32).ravel(), np.asarray(g["los"]["tau"], np.float32))
    tr.close()


def test_full_size_properties_c3():
    """Size-independent properties at the benchmark size."""
    import torch
    tr = _tracer(K.C3)
    tr.trace()
    counts = tr.counts()
    nb = tr.nb
    assert all(counts[b + 1] <= counts[b] for b in range(nb))       # rays only ever die
    keyed = []
    prev_tau = None
    for b in range(nb):
        n = int(counts[b + 1])
        h = tr.hits(b, n)
        r = tr.records(b, n)
        ray = h["ray"].to(torch.int64) & 0xFFFFFFFF
        assert torch.unique(ray).numel() == n                       # a ray hits once per bounce
        ub = r["unblocked"]
        nrm = r["dirx"] ** 2 + r["diry"] ** 2 + r["dirz"] ** 2
        assert float((nrm[ub] - 1).abs().max()) < 1e-6              # unit arrival directions
        assert bool((r["tau"][ub] > h["tau"].expand_as(r["tau"])[ub]).all())   # last leg adds delay
        blocked_ok = (r["a_te_re"][~ub] == 0) & (r["tau"][~ub] == 0)
        assert bool(blocked_ok.all())
        # delay grows from bounce to bounce for the same ray
        tau_now = torch.zeros(tr.ntx * tr.num_local, device=ray.device)
        tau_now[ray] = h["tau"]
        if prev_tau is not None:
            assert bool((tau_now[ray] > prev_tau[ray]).all())
        prev_tau = tau_now
        order = torch.argsort(ray)
        keyed.append((ray[order].clone(), h["tau"][order].clone(), r["a_te_re"][:, order].clone()))
    # a second run compacts in a different order but every record is a function of its ray
    tr.trace()
    for b in range(nb):
        n = int(counts[b + 1])
        h = tr.hits(b, n)
        r = tr.records(b, n)
        ray = h["ray"].to(torch.int64) & 0xFFFFFFFF
        order = torch.argsort(ray)
        assert torch.equal(ray[order], keyed[b][0])
        assert torch.equal(h["tau"][order].view(torch.int32), keyed[b][1].view(torch.int32))
        assert torch.equal(r["a_te_re"][:, order].view(torch.int32), keyed[b][2].view(torch.int32))
    tr.close()


def _subset_check(tr, c, stride):
    """Records of the rays p = 0, stride, 2*stride, ... of every TX, at the FULL launch-set size,
    against the oracle run on exactly those rays (oracle.compute_paths_subset: compact arrays)."""
    import torch
    from hermespy_rt_amd.abi import written
    from oracle import oracle
    sub = oracle.compute_paths_subset(*K.args(c), subset=(0, c["num_paths"], stride))
    counts = tr.counts()
    nrx, ntx, nb = tr.nrx, tr.ntx, tr.nb
    order = torch.from_numpy(tr.tri_order.astype(np.int64)).to(tr.device)
    n_amp_diff, n_cmp, worst = 0, 0, 0.0
    for b in range(nb):
        n = int(counts[b + 1])
        want = written(sub["scat"]["a_te_re"][0, :, b, :])                      # [ntx, n_sub] hit at bounce b
        if n == 0:
            assert not want.any()
            continue
        h = tr.hits(b, n)
        ray = h["ray"].to(torch.int64) & 0xFFFFFFFF
        tx, p = tr.global_path(ray)
        sel = torch.nonzero(p % stride == 0).squeeze(1)
        txs, ks = tx[sel].cpu().numpy(), (p[sel] // stride).cpu().numpy()
        got = np.zeros_like(want)
        got[txs, ks] = True
        assert np.array_equal(got, want), "bounce %d: the subset rays that hit differ from the oracle" % b
        tri = order[h["tri"].to(torch.int64)[sel] & 0xFFFFFFFF].cpu().numpy()
        assert np.array_equal(tri.astype(np.uint32), sub["hit_tri"][b, txs, ks]), "bounce %d: hit triangles" % b
        r = tr.records(b, n)
        for rx in range(nrx):
            ub = r["unblocked"][rx][sel].cpu().numpy()
            for k in ("tau", "a_te_re", "a_te_im", "a_tm_re", "a_tm_im"):
                a = r[k][rx][sel].cpu().numpy()
                e = sub["scat"][k][rx, txs, b, ks]
                if k == "tau":
                    assert np.array_equal(a.view(np.uint32), e.view(np.uint32)), "bounce %d rx %d tau" % (b, rx)
                else:
                    d = a.view(np.uint32) != e.view(np.uint32)
                    n_amp_diff += int(d.sum())
                    n_cmp += d.size
                    if d.any():
                        worst = max(worst, float(np.max(np.abs(a[d] - e[d]) / np.maximum(np.abs(e[d]), 1e-30))))
            for ci, dk in enumerate(("dirx", "diry", "dirz")):
                a = r[dk][rx][sel].cpu().numpy()[ub]
                e = sub["scat"]["directions_rx"][rx, txs, b, ks, ci][ub]
                assert np.array_equal(a.view(np.uint32), e.view(np.uint32)), "bounce %d rx %d %s" % (b, rx, dk)
            # a blocked record has zeros and leaves its direction slot untouched
            assert not written(sub["scat"]["directions_rx"][rx, txs, b, ks, 0][~ub]).any()
    # amplitudes are bit-identical: the float libm is restated bit for bit, and the device's double
    # acos of the incidence angle equals glibc's on every float input (tests/exhaustive_incidence.py)
    assert worst <= 1e-5, "amplitude relative error %.3g" % worst
    assert n_amp_diff == 0, "%d of %d amplitude words differ" % (n_amp_diff, n_cmp)
    return n_cmp, n_amp_diff


def test_c4_full_size_subset_against_oracle():
    tr = _tracer(K.C4)
    tr.trace()
    n_cmp, n_diff = _subset_check(tr, K.C4, 4099)
    assert n_cmp > 1000
    tr.close()


def test_c5_full_size_on_one_gpu():
    """BASELINE configs[4] whole (8 TX x 8 RX x 8 bounces, 64 M rays, ~190 GB of workspace) on one
    MI355X: size-independent properties, the 8 ray shards an 8-GPU run would hold add up to the
    whole, and a strided subset of the rays against the oracle at the full launch-set size."""
    import torch
    c = K.C5
    tr = _tracer(c)
    tr.trace()
    counts = tr.counts()
    nb = tr.nb
    live = [int(x) for x in counts[:nb + 1]]
    assert live[0] == 8 * 8000000 and all(live[b + 1] <= live[b] for b in range(nb)) and live[nb] > 0
    for b in (0, nb - 1):
        n = int(counts[b + 1])
        ray = tr.hits(b, n)["ray"].to(torch.int64) & 0xFFFFFFFF
        assert int(torch.unique(ray).numel()) == n                       # a ray hits once per bounce
        del ray
    n_cmp, n_diff = _subset_check(tr, c, 4099)
    assert n_cmp > 100000
    tr.close()
    del tr
    torch.cuda.empty_cache()
    total = np.zeros(nb + 1, np.int64)
    for r in range(8):   # what each GPU of the 8-GPU run holds
        ts = _tracer(c, rank=r, world=8)
        ts.trace()
        total += np.asarray(ts.counts()[:nb + 1], np.int64)
        ts.close()
        del ts
        torch.cuda.empty_cache()
    assert [int(x) for x in total] == live
