#!/usr/bin/env python3
"""One traced step on the generated city (for rocprofv3 --pmc passes): python3 profiles/pmc_city.py N_SIDE [RAYS]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hermespy_rt_amd.device import Tracer  # noqa: E402
from tests import scenes_gen as G  # noqa: E402

n = int(sys.argv[1])
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
p = os.path.join(tempfile.mkdtemp(), "city.hrt")
G.city(p, n)
tr = Tracer(p, [[60.0, 0.0, 1.5], [0.0, -90.0, 1.5], [-150.0, 30.0, 1.5]], [[0.0, 0.0, 25.0]], [[0, 0, 0]] * 3,
            [[0, 0, 0]], 3.5, rays, 2)
tr.trace()
torch.cuda.synchronize()
tr.trace()
torch.cuda.synchronize()
