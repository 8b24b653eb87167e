"""The three intersection loops of the bounce kernel -- HRT_TUNE variant=0 (the reference's
plain sequence), 1 (staged division-free rejects), 2 (packet culling + staged; the default)
-- must give bit-identical results.  The variant is latched per process, so each one runs in
a subprocess through the drop-in C ABI against the oracle, on a coherent and an incoherent
launch order."""
import os
import subprocess
import sys

import pytest

from tests.tune import tuned

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys
sys.path.insert(0, %(repo)r)
import numpy as np
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K
from tests.parity import compare_dense
for c in (K.small(K.C3, 30000), K.small(K.C4_DOPPLER, 7000), K.small(K.C5, 2048)):
    got = abi.run_compute_paths(lib.load(), *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for k, v in st.items()), st
# device API with and without the coherent launch order
from hermespy_rt_amd.device import Tracer
c = K.small(K.C3, 50000)
outs = []
for coh in (True, False):
    tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"],
                c["num_paths"], c["num_bounces"], coherent=coh)
    tr.trace()
    outs.append(tr.to_dense())
    tr.close()
for k in outs[0]:
    assert np.array_equal(outs[0][k].view(np.uint32) if outs[0][k].dtype == np.float32 else outs[0][k],
                          outs[1][k].view(np.uint32) if outs[1][k].dtype == np.float32 else outs[1][k]), k
print("VARIANT_OK")
"""


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_variant_is_bit_identical(variant):
    env = tuned(variant=variant)
    p = subprocess.run([sys.executable, "-c", CODE % dict(repo=REPO)], env=env, capture_output=True, text=True)
    assert p.returncode == 0 and "VARIANT_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
