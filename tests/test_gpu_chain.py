"""hrt_chain_kernel: the tail of the launches as ONE persistent kernel (grid barriers between the bounces, the
live lists handed over with sc1 accesses; csrc/hrt_kernels.hip).  Same bits as a kernel per launch and as the
oracle -- on lists that stay long through every bounce (the closed box: nothing leaves, every workgroup loops
over several chunks per bounce) and on lists that run empty half way (2cars), from every first launch."""
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
L = lib.load()
cases = [dict(K.small(K.C1, %(box_rays)d), num_bounces=%(nb)d), dict(K.small(K.C4_DOPPLER, 200000), num_bounces=%(nb)d),
         dict(K.small(K.C2, 100000), num_bounces=%(nb)d)]
for c in cases:
    got = abi.run_compute_paths(L, *K.args(c))
    st = compare_dense(got, oracle.compute_paths(*K.args(c)))
    assert all(v == 0 for v in st.values()), (c["scene_path"], st)
print("CHAIN_OK")
"""


@pytest.mark.parametrize("chain_from,nb,box_rays", [(1, 4, 700000), (2, 4, 300000), (2, 7, 40000), (3, 5, 40000), (None, 6, 300000)])
def test_chain_kernel_equals_the_oracle(chain_from, nb, box_rays):
    env = tuned() if chain_from is None else tuned(chain_from=chain_from)
    p = subprocess.run([sys.executable, "-c", CODE % dict(repo=REPO, nb=nb, box_rays=box_rays)], env=env,
                       capture_output=True, text=True)
    assert p.returncode == 0 and "CHAIN_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_chain_and_one_kernel_per_launch_leave_the_same_workspace():
    """device API: every block of the compact result (hit lists, records, masks, counts), chain against no chain"""
    code = r"""
import sys
sys.path.insert(0, %(repo)r)
import numpy as np
from hermespy_rt_amd.device import Tracer
from tests import configs as K
c = dict(K.small(K.C1, 600000), num_bounces=5)
tr = Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], c["f_ghz"], c["num_paths"], c["num_bounces"])
tr.trace()
cnt = tr.counts()
out = dict(counts=cnt)
for b in range(c["num_bounces"]):
    h = int(cnt[b + 1])
    out["hit%%d" %% b] = tr.hit_block(b)[:, :h].cpu().numpy().copy()
    out["rec%%d" %% b] = tr.rec_block(b)[:, :, :h].cpu().numpy().copy()
    out["mask%%d" %% b] = tr.mask_block(b)[:, :2 * ((h + 63) // 64)].cpu().numpy().copy()
np.savez(sys.argv[1], **out)
assert cnt[c["num_bounces"]] > 300000   # (the box is closed: the lists stay long)
tr.close()
"""
    import tempfile

    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for name, env in (("chain", tuned(chain_from=1)), ("plain", tuned(no_chain=1))):
            f = os.path.join(d, name + ".npz")
            p = subprocess.run([sys.executable, "-c", code % dict(repo=REPO), f], env=env, capture_output=True, text=True)
            assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
            outs.append(np.load(f))
        assert sorted(outs[0].files) == sorted(outs[1].files)
        nb = sum(1 for k in outs[0].files if k.startswith("hit"))
        for k in outs[0].files:
            a, b = outs[0][k], outs[1][k]
            assert a.shape == b.shape, k
            if k.startswith("rec"):
                # (a blocked record is five zeros -- amplitudes and delay; its other four words are not written)
                assert np.array_equal(a[:, :5].view(np.uint32), b[:, :5].view(np.uint32)), k
                m = outs[0]["mask" + k[3:]].view(np.uint32)
                h = a.shape[2]
                bits = np.unpackbits(m.view(np.uint8), axis=1, bitorder="little")[:, :h].astype(bool)
                for rx in range(a.shape[0]):
                    assert np.array_equal(a[rx][:, bits[rx]].view(np.uint32), b[rx][:, bits[rx]].view(np.uint32)), (k, rx)
            else:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
        assert nb == 5


CODE_BIG = r"""
import sys
sys.path.insert(0, %(repo)r)
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K
from tests.parity import compare_dense
L = lib.load()
c = dict(K.small(K.C3_DOPPLER, 120000), num_bounces=5)
st = compare_dense(abi.run_compute_paths(L, *K.args(c)), oracle.compute_paths(*K.args(c)))
assert all(v == 0 for v in st.values()), st
print("CHAIN_OK")
"""


@pytest.mark.parametrize("variant", [2, 4])
def test_chain_kernel_on_a_table_of_one_culling_block(variant):
    """HRT_FUSE=2 fuses whole bounces on any table of one culling block (<= 1 024 triangles): the chain kernel with
    the packet-culling walks (234 triangles in LDS), flat and behind the leaf spheres"""
    env = tuned(HRT_FUSE=2, no_patch=1, chain_from=1, variant=variant)
    p = subprocess.run([sys.executable, "-c", CODE_BIG % dict(repo=REPO)], env=env, capture_output=True, text=True)
    assert p.returncode == 0 and "CHAIN_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
