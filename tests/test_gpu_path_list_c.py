"""hrt_compute_paths_list (C ABI): compute_paths with the result as one list of records.  Every
entry must carry the oracle's dense value at its slot, there is one entry per non-zero record (or
per written record with include_blocked), mesh/face name the oracle's hit triangle, and the LoS
block equals the dense LoS outputs."""
import numpy as np
import pytest

from hermespy_rt_amd import abi, lib
from oracle import oracle

from . import configs as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,npaths", [("C3", 20000), ("C4_DOPPLER", 9001), ("C1", 10000), ("C5", 1500)])
def test_c_path_list_equals_dense_oracle(product_lib, name, npaths):
    c = K.small(K.ALL[name], npaths)
    ref = oracle.compute_paths(*K.args(c))
    sc = ref["scat"]
    written = abi.written(sc["a_te_re"])
    unblocked = abi.written(sc["directions_rx"][..., 0])
    ht = np.asarray(ref["extras"]["hit_tri"])
    mesh_ids, face_ids = np.asarray(ref["extras"]["tri_mesh"]), np.asarray(ref["extras"]["tri_face"])
    for include_blocked in (False, True):
        st = lib.Stats()
        P = abi.run_compute_paths_list(product_lib, *K.args(c), include_blocked=include_blocked, stats=st)
        want = written if include_blocked else unblocked
        n = P["rx"].size
        assert n == int(want.sum())
        idx = (P["rx"].astype(np.int64), P["tx"].astype(np.int64), P["bounce"].astype(np.int64),
               P["path"].astype(np.int64))
        assert np.unique(np.ravel_multi_index(idx, written.shape)).size == n and want[idx].all()
        for k in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau"):
            assert np.array_equal(P[k].view(np.uint32), sc[k][idx].view(np.uint32)), k
        ub = P["unblocked"]
        assert np.array_equal(ub, unblocked[idx])
        assert np.array_equal(P["direction_rx"][ub].view(np.uint32), sc["directions_rx"][idx][ub].view(np.uint32))
        if len(c["tx_pos"]) == 1:
            assert np.array_equal(P["freq_shift"][ub], sc["freq_shift"][idx][ub])
        t = ht[idx[2], idx[1], idx[3]]
        assert np.array_equal(P["mesh"], mesh_ids[t]) and np.array_equal(P["face"], face_ids[t])
        assert int(st.records_unblocked) == int(unblocked.sum()) and int(st.records) == int(written.sum())
        # LoS block: status 2 = clear -> the dense LoS values
        los = P["los"]
        status = los[..., 0].view(np.uint32)
        clear = status == 2
        assert np.array_equal(los[..., 1][clear], ref["los"]["a_te_re"][clear])
        assert np.array_equal(los[..., 2][clear], ref["los"]["tau"][clear])


def test_more_host_threads_than_the_writers_have_slots(product_lib, monkeypatch):
    """HRT_HOST_THREADS above 32 used to index past the per-thread arrays of the list
    writer once a hit block had more than 2.1 M entries (C3 at 3 M rays has 2.4 M at bounce 0).
    The value is clamped now; the list and the dense form must come out the same as with 16."""
    c = K.small(K.C3, 3000000)
    monkeypatch.setenv("HRT_HOST_THREADS", "16")
    st16 = lib.Stats()
    a = abi.run_compute_paths_list(product_lib, *K.args(c), stats=st16)
    monkeypatch.setenv("HRT_HOST_THREADS", "64")
    st64 = lib.Stats()
    b = abi.run_compute_paths_list(product_lib, *K.args(c), stats=st64)
    assert int(st64.live[1]) > 2200000
    assert a["rx"].size == b["rx"].size and int(st16.records) == int(st64.records)
    for k in ("rx", "tx", "bounce", "path", "a_te_re", "tau", "mesh", "face"):
        assert np.array_equal(a[k], b[k]), k
    d64 = abi.run_compute_paths(product_lib, *K.args(K.small(K.C3, 600000)), with_rays=False)
    monkeypatch.setenv("HRT_HOST_THREADS", "16")
    d16 = abi.run_compute_paths(product_lib, *K.args(K.small(K.C3, 600000)), with_rays=False)
    for k in ("a_te_re", "tau", "freq_shift"):
        assert np.array_equal(d64["scat"][k].view(np.uint32), d16["scat"][k].view(np.uint32)), k


def test_list_of_several_batches_and_reused_blocks(product_lib, monkeypatch):
    """The list of a call that runs as several batches (its size comes from the first batch, it must not be
    short) equals the one-batch list as a set of records; and a list written into the blocks kept from the
    last freed one (hrt_path_list_free keeps one set) carries nothing of the old list."""
    c = K.small(K.C4_DOPPLER, 300000)
    one = abi.run_compute_paths_list(product_lib, *K.args(c), include_blocked=True)
    small = abi.run_compute_paths_list(product_lib, *K.args(K.small(K.C2, 20000)), include_blocked=True)   # into kept blocks
    assert small["rx"].size > 0
    monkeypatch.setenv("HRT_WORKSPACE_BYTES", str(60 << 20))
    st = lib.Stats()
    many = abi.run_compute_paths_list(product_lib, *K.args(c), include_blocked=True, stats=st)
    assert int(st.num_batches) >= 4
    assert many["rx"].size == one["rx"].size

    def key(P):
        k = ((P["rx"].astype(np.int64) * 4 + P["tx"]) * 64 + P["bounce"]) * (1 << 32) + P["path"].astype(np.int64)
        return np.argsort(k, kind="stable"), k
    oa, ka = key(one)
    ob, kb = key(many)
    assert np.array_equal(ka[oa], kb[ob])
    for f in ("a_te_re", "a_te_im", "a_tm_re", "a_tm_im", "tau", "unblocked", "mesh", "face"):
        assert np.array_equal(one[f][oa].view(np.uint8), many[f][ob].view(np.uint8)), f
    # (a blocked record has no direction and no Doppler term: the reference does not write those slots, Q2)
    ub = one["unblocked"][oa].astype(bool)
    assert np.array_equal(one["direction_rx"][oa][ub].view(np.uint32), many["direction_rx"][ob][ub].view(np.uint32))
    assert np.array_equal(one["freq_shift"][oa][ub].view(np.uint32), many["freq_shift"][ob][ub].view(np.uint32))
