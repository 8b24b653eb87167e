"""The drop-in Python module `hermespy_rt` (pybind11): the reference's own smoke test
(test/test.py: simple_reflector.hrt, 10 000 rays, 3 bounces; float64 inputs; shape asserts)
plus values against the oracle."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle

from . import configs as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    import hermespy_rt_amd
    import torch  # noqa: F401  (HIP runtime first, see hermespy_rt_amd.lib)
    sys.path.insert(0, hermespy_rt_amd.LIB_DIR)
    import hermespy_rt
    return hermespy_rt


def test_reference_smoke_test_contract(rt):
    c = K.TEST_PY
    num_rx = num_tx = 1
    num_paths, num_bounces = 10000, 3
    los, scatter = rt.compute_paths(
        c["scene_path"], np.array(c["rx_pos"], dtype=np.float64), np.array(c["tx_pos"], dtype=np.float64),
        np.array(c["rx_vel"], dtype=np.float64), np.array(c["tx_vel"], dtype=np.float64),
        3.0, num_rx, num_tx, num_paths, num_bounces)
    # test/test.py:61-87
    assert los.num_paths == 1
    assert (num_rx, num_tx, 1, 3) == los.directions_rx.shape == los.directions_tx.shape
    assert (num_rx, num_tx, 1) == los.a_te.shape == los.a_tm.shape == los.tau.shape == los.freq_shift.shape
    assert scatter.num_paths == num_bounces * num_paths
    assert (num_rx, num_tx, scatter.num_paths, 3) == scatter.directions_rx.shape == scatter.directions_tx.shape
    assert (num_rx, num_tx, scatter.num_paths) == scatter.a_te.shape == scatter.a_tm.shape \
        == scatter.tau.shape == scatter.freq_shift.shape
    assert los.a_te.dtype == np.complex64 and scatter.tau.dtype == np.float32
    # values printed by the reference for this input
    assert np.float32(los.tau[0, 0, 0]) == np.float32(3.335598e-12) or abs(los.tau[0, 0, 0] - 3.335598e-12) < 1e-18
    assert los.a_te[0, 0, 0] == 1 + 0j
    # against the oracle: written slots carry the reference's values, the rest reads 0
    ref = oracle.compute_paths(*K.args(c))
    w = ref["scat"]["a_te_re"].view(np.uint32) != oracle.SENTINEL_U32
    assert w.sum() == 3690                                   # hits 3690 of 10000, then 0
    tau_ref = np.where(w, ref["scat"]["tau"], 0).reshape(1, 1, -1)
    assert np.array_equal(scatter.tau.view(np.uint32), tau_ref.astype(np.float32).view(np.uint32))
    a_ref = np.where(w, ref["scat"]["a_te_re"] + 1j * ref["scat"]["a_te_im"], 0).reshape(1, 1, -1)
    assert np.allclose(scatter.a_te, a_ref, rtol=1e-5, atol=0)
    assert not scatter.directions_tx.any()                   # never written (Q1) -> zeros here


def test_keyword_names_and_errors(rt):
    c = K.small(K.C1, 100)
    kw = dict(mesh_filepath=c["scene_path"], rx_positions=np.array(c["rx_pos"], np.float32),
              tx_positions=np.array(c["tx_pos"], np.float32), rx_velocities=np.zeros((1, 3), np.float32),
              tx_velocities=np.zeros((1, 3), np.float32), carrier_frequency=3.0, num_rx=1, num_tx=1,
              num_paths=100, num_bounces=1)
    los, sc = rt.compute_paths(**kw)
    assert sc.a_te.shape == (1, 1, 100)
    with pytest.raises(ValueError):
        rt.compute_paths(**dict(kw, mesh_filepath="/nonexistent.hrt"))
    with pytest.raises(Exception):
        rt.compute_paths(**dict(kw, num_paths=0))


def test_compute_paths_list_matches_the_dense_result(rt):
    """The list extension of the module: every entry equals the dense output at its slot
    [rx, tx, bounce * num_paths + path]; one entry per non-zero record."""
    c = K.small(K.C3, 6000)
    nrx, ntx, npaths, nb = len(c["rx_pos"]), len(c["tx_pos"]), c["num_paths"], c["num_bounces"]
    args = (c["scene_path"], np.array(c["rx_pos"], np.float32), np.array(c["tx_pos"], np.float32),
            np.array(c["rx_vel"], np.float32), np.array(c["tx_vel"], np.float32), c["f_ghz"], nrx, ntx,
            npaths, nb)
    los, scatter = rt.compute_paths(*args)
    P = rt.compute_paths_list(*args)
    n = P["rx"].size
    slot = (P["rx"].astype(np.int64), P["tx"].astype(np.int64),
            P["bounce"].astype(np.int64) * npaths + P["path"].astype(np.int64))
    assert n == int((scatter.directions_rx != 0).any(axis=-1).sum()) and P["unblocked"].all()
    assert P["a_te"].dtype == np.complex64 and P["path"].dtype == np.uint64
    assert np.array_equal(P["a_te"], scatter.a_te[slot]) and np.array_equal(P["a_tm"], scatter.a_tm[slot])
    assert np.array_equal(P["tau"], scatter.tau[slot])
    assert np.array_equal(P["direction_rx"], scatter.directions_rx[slot])
    assert np.array_equal(P["freq_shift"], scatter.freq_shift[slot])
    assert P["los"].shape == (nrx, ntx, 8)
    clear = P["los"][..., 0].view(np.uint32) == 2
    assert np.array_equal(P["los"][..., 2][clear], los.tau[..., 0][clear])
    both = rt.compute_paths_list(*args, include_blocked=True)
    # blocked records (zeros in the dense form) are listed too, flagged
    assert both["rx"].size > n and int(both["unblocked"].sum()) == n
    blk = ~both["unblocked"]
    assert not both["a_te"][blk].any() and not both["tau"][blk].any()


@pytest.mark.parametrize("name", ["C3_20k", "C4_DOPPLER_5k", "C3_20k_two_devices"])
def test_complex_amplitudes_written_in_place(rt, product_lib, name, monkeypatch):
    """The module hands its complex64 arrays to hrt_compute_paths_interleaved (re at [2 i], im at
    [2 i + 1]); the reference's binding fills four planes and interleaves them
    (compute_paths_pybind11.cpp:44-97).  Same bits as the planes of hrt_compute_paths_ex, LoS
    block included; slots nobody writes read 0."""
    from hermespy_rt_amd import abi
    if name.endswith("two_devices"):
        monkeypatch.setenv("HRT_DEVICES", "0,0")   # two logical devices write into the same complex arrays
    c = K.small(K.C3, 20000) if name.startswith("C3_20k") else K.small(K.C4_DOPPLER, 5000)
    f32 = lambda a: np.array(a, np.float32)   # noqa: E731
    los, scat = rt.compute_paths(c["scene_path"], f32(c["rx_pos"]), f32(c["tx_pos"]), f32(c["rx_vel"]),
                                 f32(c["tx_vel"]), c["f_ghz"], len(c["rx_pos"]), len(c["tx_pos"]),
                                 c["num_paths"], c["num_bounces"])
    ref = abi.run_compute_paths(product_lib, *K.args(c))
    for blk, got in (("los", los), ("scat", scat)):
        for pol in ("a_te", "a_tm"):
            w = abi.written(ref[blk][pol + "_re"])
            re = np.where(w, ref[blk][pol + "_re"], np.float32(0)).astype(np.float32)
            im = np.where(abi.written(ref[blk][pol + "_im"]), ref[blk][pol + "_im"], np.float32(0)).astype(np.float32)
            g = np.ascontiguousarray(getattr(got, pol)).reshape(-1)
            assert g.dtype == np.complex64
            assert np.array_equal(g.real.view(np.uint32), re.reshape(-1).view(np.uint32)), (blk, pol, "re")
            assert np.array_equal(g.imag.view(np.uint32), im.reshape(-1).view(np.uint32)), (blk, pol, "im")
        tau = np.where(abi.written(ref[blk]["tau"]), ref[blk]["tau"], np.float32(0)).astype(np.float32)
        assert np.array_equal(np.ascontiguousarray(got.tau).reshape(-1).view(np.uint32), tau.reshape(-1).view(np.uint32))


def test_cache_clear_is_exposed_and_calls_still_work(rt):
    """hermespy_rt.cache_clear() releases the buffers kept between calls; the next
    call allocates again and returns the same values."""
    c = K.small(K.C1, 2000)
    args = (c["scene_path"], np.array(c["rx_pos"], np.float32), np.array(c["tx_pos"], np.float32),
            np.array(c["rx_vel"], np.float32), np.array(c["tx_vel"], np.float32), c["f_ghz"], 1, 1, 2000, 1)
    a = rt.compute_paths(*args)[1]
    rt.cache_clear()
    rt.cache_clear()          # nothing kept: a no-op
    b = rt.compute_paths(*args)[1]
    assert np.array_equal(a.tau.view(np.uint32), b.tau.view(np.uint32))
    assert np.array_equal(a.a_te.view(np.uint32), b.a_te.view(np.uint32))
