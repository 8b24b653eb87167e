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
