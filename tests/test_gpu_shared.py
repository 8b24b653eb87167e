"""Several tracer processes on ONE GPU at the same time -- how a drop-in library is used (worker processes,
or next to torch).  The fused kernels' workgroups wait for the chunks in front of them; with chunks numbered
by dispatch (blockIdx) four such processes locked each other out for whole time slices (C4: 5.8 s per step
instead of 0.2 ms, profiles/HISTORY.md r4); now they share the GPU like any other kernels.  Default settings (HRT_FUSE unset)."""
# (what makes it safe: a fused launch gives up waiting after ~10 ms, declares the step void and the library goes
# on with two kernels per launch -- hrt_kernels.hip lb_exclusive, problem.c fuse_mode; tickets drawn at workgroup
# start were measured too: exact, but 10 % of C4's step)
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,rays,procs", [("c4", 1000000, 4), ("c3", 500000, 2)])
def test_processes_sharing_one_gpu_stay_work_conserving(workload, rays, procs):
    sys.path.insert(0, os.path.join(REPO, "profiles", "tools"))
    import shared_gpu
    solo = shared_gpu.run(workload, rays, 1, 50)[0]
    both = shared_gpu.run(workload, rays, procs, 50)
    # sharing costs each process at most its share of the GPU (x procs), with a factor 2 of slack
    assert max(both) <= 2.0 * procs * solo + 1.0, (solo, both)


def test_a_fused_launch_that_gives_up_is_redone_unfused():
    """The fallback itself, deterministically: with lb_max_polls=0 every chunk that has to wait at all declares
    the step void at once.  The drop-in call must notice (HRT_ERR_FUSE_TIMEOUT in the counts it reads), run the
    step again as two kernels per launch, keep fusion off -- and return the oracle's bits."""
    import subprocess
    code = r"""
import sys
sys.path.insert(0, %r)
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K
from tests.parity import compare_dense
L = lib.load()
for c in (K.small(K.C4_DOPPLER, 300000), K.small(K.C3, 200000), K.small(K.C4_DOPPLER, 300000)):
    st = compare_dense(abi.run_compute_paths(L, *K.args(c)), oracle.compute_paths(*K.args(c)))
    assert all(v == 0 for v in st.values()), st
print("FALLBACK_OK")
""" % REPO
    from tests.tune import tuned
    p = subprocess.run([sys.executable, "-c", code], env=tuned(lb_max_polls=0), capture_output=True, text=True)
    assert p.returncode == 0 and "FALLBACK_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
