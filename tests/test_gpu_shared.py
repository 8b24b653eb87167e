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
