"""Tracer.paths(): the resolved scatter paths as one COO list (rx, tx, bounce, path + values).
Every listed path must carry exactly the oracle's dense value at its slot, the list must have
one entry per non-zero record (or per written record with nonzero_only=False), and mesh/face
must name the triangle the oracle says the ray hit."""
import numpy as np
import pytest

from hermespy_rt_amd import abi
from oracle import oracle

from . import configs as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,npaths", [("C3", 20000), ("C4_DOPPLER", 9000), ("C1", 10000)])
def test_path_list_equals_dense_oracle(name, npaths):
    from hermespy_rt_amd.device import Tracer
    c = K.small(K.ALL[name], npaths)
    ref = oracle.compute_paths(*K.args(c))
    tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                c["num_paths"], c["num_bounces"])
    tr.trace()
    sc = ref["scat"]
    written = abi.written(sc["a_te_re"])
    unblocked = abi.written(sc["directions_rx"][..., 0])
    for nonzero_only in (True, False):
        P = {k: v.cpu().numpy() for k, v in tr.paths(nonzero_only=nonzero_only, with_geometry=True).items()}
        n = P["rx"].size
        assert n == int((unblocked if nonzero_only else written).sum())
        idx = (P["rx"], P["tx"], P["bounce"], P["path"])
        # one entry per slot
        flat = np.ravel_multi_index(idx, written.shape)
        assert np.unique(flat).size == n
        assert (unblocked if nonzero_only else written)[idx].all()
        for k, (re, im) in {"a_te": ("a_te_re", "a_te_im"), "a_tm": ("a_tm_re", "a_tm_im")}.items():
            assert np.array_equal(P[k].real.view(np.uint32), sc[re][idx].view(np.uint32)), k
            assert np.array_equal(P[k].imag.view(np.uint32), sc[im][idx].view(np.uint32)), k
        assert np.array_equal(P["tau"].view(np.uint32), sc["tau"][idx].view(np.uint32))
        ub = P["unblocked"]
        assert np.array_equal(ub, unblocked[idx])
        assert np.array_equal(P["direction_rx"][ub].view(np.uint32), sc["directions_rx"][idx][ub].view(np.uint32))
        if len(c["tx_pos"]) == 1:   # the dense freq_shift is only defined for one TX (Q9)
            assert np.array_equal(P["freq_shift"][ub], sc["freq_shift"][idx][ub])
        # geometry: the triangle the ray left towards the RX = the oracle's hit of that bounce
        ht = np.asarray(ref["extras"]["hit_tri"])          # [nb, ntx, np] flat triangle index
        mesh_ids, face_ids = np.asarray(ref["extras"]["tri_mesh"]), np.asarray(ref["extras"]["tri_face"])
        t = ht[P["bounce"], P["tx"], P["path"]]
        assert np.array_equal(P["mesh"], mesh_ids[t]) and np.array_equal(P["face"], face_ids[t])
    tr.close()
