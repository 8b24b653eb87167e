"""The acceleration structure (csrc/host/accel.c, closest_hit_tree / closest_hit_big
in csrc/hrt_kernels.hip; reference: the "TODO BVH" of src/compute_paths.c:246, semantics :237-287).

Whatever the structure skips must be exactly what the reference's float test rejects, including
its noise-regime hits on triangles whose plane contains the ray -- so every case is the product
against the oracle, every output array bit for bit:
  * the modes: leaf spheres + guard (variant=4), inner levels + plane tree
    forced onto small tables (accel_big=0: every scene of the suite then walks the trees),
    the reference's own order (no_reorder=1), the flat walk (2) -- on generated scenes,
    exact ties (duplicated triangles: the lexicographic (distance, original index) tie-break),
    endpoints exactly IN triangle planes, degenerate triangles;
  * a generated city of 10^5 triangles (tests/scenes_gen.city): default, trees, plain flat walk;
  * the re-sort of the live list between bounces (sort_rays) on and off.
The variant and the table order are latched per process / per problem: subprocesses."""
import os
import subprocess
import sys

import pytest

from tests.tune import tuned

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, tempfile
sys.path.insert(0, %(repo)r)
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K
from tests.parity import compare_dense
from tests.test_generated_scenes import make, NAMES
tmp = tempfile.mkdtemp()
cases = [make(tmp, n) for n in NAMES] + [K.small(K.C3, 20000), K.small(K.C4_DOPPLER, 5000), K.small(K.C5, 1024)]
cases += list(K.IN_PLANE.values())
for c in cases:
    got = abi.run_compute_paths(lib.load(), *K.args(c))
    ref = oracle.compute_paths(*K.args(c))
    st = compare_dense(got, ref)
    assert all(v == 0 for v in st.values()), (c["scene_path"], st)
print("ACCEL_OK", len(cases))
"""


@pytest.mark.parametrize("env", [
    dict(variant="4"),                   # leaf spheres + guard
    dict(accel_big="0"),                       # inner levels + plane tree on every table (auto picks them)
    dict(accel_big="0", no_reorder="1"),   # ... on the reference's own table order
    dict(variant="2"),                   # flat packet culling on the reordered table
    dict(sort_rays="1"),                       # live list re-sorted between bounces on every table
    dict(sort_rays="0"),                       # ... and never
    dict(accel_big="0", sort_rays="1"),    # trees + re-sort
    dict(variant="0", no_reorder="1"),
    dict(no_txt="1"),                          # direction tables for the RXs only (default: RXs and TXs)
    dict(no_rxt="1"),                          # ... and none at all
    dict(rxt_min_rays="67108864"),             # the drop-in's own default: tables from 2^26 rays on (none here)
    # fine leaves (16 rows) scanned flat + plane tree (the default beyond 1 024 triangles), forced onto every
    # table, the table read from global memory as there
    dict(accel_fine_min="0", lds_tri_bytes="0"),
    dict(accel_fine_min="0", lds_tri_bytes="0", sort_rays="1"),
    dict(accel_fine_min="0", lds_tri_bytes="0", no_reorder="1"),
    # ... its queue of too-wide packets (hrt_wide_kernel) overflowing after 3 entries (the rest runs in the
    # pushing wave), and absent
    dict(accel_fine_min="0", lds_tri_bytes="0", wide_cap="3"),
    dict(accel_fine_min="0", lds_tri_bytes="0", wide_cap="0", sort_rays="1"),
    dict(los_big_min_tri="0"),                 # the big tables' sliced LoS kernel on every table
    # fine leaves + queue on two logical devices, each running several batches through a small workspace
    dict(accel_fine_min="0", lds_tri_bytes="0", HRT_DEVICES="0,0", HRT_WORKSPACE_BYTES="8000000"),
], ids=["leaf", "trees", "trees_ref_order", "flat", "resort", "no_resort", "trees_resort", "plain_ref_order",
        "rx_tables_only", "no_tables", "tables_by_size", "fine", "fine_resort", "fine_ref_order", "fine_queue_overflow",
        "fine_no_queue", "los_sliced", "fine_devices_batches"])
def test_modes_are_bit_identical_to_the_oracle(env):
    p = subprocess.run([sys.executable, "-c", CODE % dict(repo=REPO)], env=tuned(**env),
                       capture_output=True, text=True)
    assert p.returncode == 0 and "ACCEL_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


CITY = r"""
import sys, tempfile, os, time
sys.path.insert(0, %(repo)r)
import numpy as np
from hermespy_rt_amd import abi, lib
from oracle import oracle
from tests import configs as K, scenes_gen as G
from tests.parity import compare_dense
p = os.path.join(tempfile.mkdtemp(), "city.hrt")
T, half = G.city(p, 100, moving=True)
assert T == 100002
c = G.cfg(p, [[60.0, 0.0, 1.5], [0.0, -90.0, 1.5], [-150.0, 30.0, 1.5]], [[0.0, 0.0, 25.0]], 6000, 3,
          rx_vel=[[1, 2, 0], [0, -3, 1], [2, 0, 0]], tx_vel=[[10, 0, 0]])
t0 = time.time()
got = abi.run_compute_paths(lib.load(), *K.args(c))
t1 = time.time()
ref = oracle.compute_paths(*K.args(c))
st = compare_dense(got, ref)
assert all(v == 0 for v in st.values()), st
live = [int(x) for x in ref["extras"]["live"]]
assert live[1] > 3000 and live[3] > 500, live
print("CITY_OK", live, "product %%.1f s, oracle %%.1f s" %% (t1 - t0, time.time() - t1))
"""


@pytest.mark.parametrize("env", [dict(), dict(wide_cap="40"), dict(accel_fine="0"), dict(accel_big="65536"),
                                 dict(variant="2", sort_rays="0")],
                         ids=["default_fine_resorted", "fine_queue_overflow", "leaves_resorted", "trees_resorted", "flat"])
def test_city_of_1e5_triangles(env):
    """10^5 triangles: by default the fine leaves (16 rows, flat scan) + plane tree over a live list
    re-sorted between bounces; the leaf spheres of 64 rows + guard (round 2's default); with
    accel_big lowered the sphere levels + plane tree; and the plain flat walk."""
    p = subprocess.run([sys.executable, "-c", CITY % dict(repo=REPO)], env=tuned(**env),
                       capture_output=True, text=True)
    assert p.returncode == 0 and "CITY_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
