"""CPU sanitizer builds (GPU AddressSanitizer is not available on the pool):
  * the product's host-only C -- scene_io.c, sionna_import.c, materials.c, accel.c -- under
    ASan + UBSan (tests/asan/), fed valid, truncated and corrupted .hrt / PLY / CSV / XML files and
    random triangle tables with NaNs, zero-area and duplicated triangles;
  * the oracle's C restatement under ASan + UBSan (make -C oracle asan), run from Python with the
    sanitizer runtime preloaded, against a golden fixture;
  * the parallel-for of the host writers (csrc/host/parallel.c: parked helper threads, one pool per
    calling thread) under ThreadSanitizer and under ASan + UBSan.
A sanitizer report ends the process with a status other than 0 (ok) or 8 (the loader's own
"bad file" exit, the reference's code, src/scene.c:36-83)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_DIR = os.path.join(REPO, "tests", "asan")
EXE = os.path.join(ASAN_DIR, "host_asan")
FIX = os.path.join(REPO, "tests", "golden", "sionna_fixture")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")


@pytest.fixture(scope="module")
def exe():
    p = subprocess.run(["make", "-C", ASAN_DIR], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return EXE


def run(exe, *args, ok=(0,)):
    p = subprocess.run([exe, *[str(a) for a in args]], env=ENV, capture_output=True, text=True, timeout=300)
    assert p.returncode in ok, "%s -> %d\n%s" % (args, p.returncode, p.stderr[-3000:])
    assert "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    return p.returncode


def test_bundled_scenes_load_and_round_trip(exe, tmp_path):
    for name in os.listdir(os.path.join(REPO, "scenes")):
        if name.endswith(".hrt"):
            q = tmp_path / name
            shutil.copy(os.path.join(REPO, "scenes", name), q)
            run(exe, "load", q)


def test_truncated_and_corrupted_hrt(exe, tmp_path):
    src = open(os.path.join(REPO, "scenes", "2cars.hrt"), "rb").read()
    rng = np.random.default_rng(5)
    cases = [src[:n] for n in (0, 2, 3, 6, 7, 11, 40, len(src) // 2, len(src) - 1)]
    for _ in range(40):                      # random byte flips, mostly in the headers / index arrays
        b = bytearray(src)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, min(len(b), 400)))] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    huge = bytearray(src)
    huge[7:11] = (0xFFFFFFF0).to_bytes(4, "little")      # first mesh: 4 G vertices
    cases.append(bytes(huge))
    rcs = set()
    for k, data in enumerate(cases):
        q = tmp_path / ("c%d.hrt" % k)
        q.write_bytes(data)
        rcs.add(run(exe, "load", q, ok=(0, 8)))
    assert 8 in rcs                          # the loader refuses, with the reference's exit code


def test_sionna_importer_on_good_and_broken_inputs(exe, tmp_path):
    run(exe, "sionna", os.path.join(FIX, "scene.xml"))
    rng = np.random.default_rng(9)
    for k in range(30):
        d = tmp_path / ("s%d" % k)
        shutil.copytree(FIX, d)
        files = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if not f.endswith(".hrt")]
        victim = files[int(rng.integers(0, len(files)))]
        data = bytearray(open(victim, "rb").read())
        mode = k % 3
        if mode == 0 and len(data) > 4:
            data = data[: int(rng.integers(0, len(data)))]                    # truncation
        elif mode == 1 and len(data):
            for _ in range(8):
                data[int(rng.integers(0, len(data)))] = int(rng.integers(0, 256))   # byte flips
        else:
            data = data + bytes(rng.integers(0, 256, 64, dtype=np.uint8))     # trailing garbage
        open(victim, "wb").write(bytes(data))
        run(exe, "sionna", os.path.join(d, "scene.xml"))
    run(exe, "sionna", tmp_path / "does_not_exist.xml")
    run(exe, "sionna", os.path.join(REPO, "scenes", "box.hrt"))               # not an .xml name


@pytest.mark.parametrize("T,seed", [(0, 1), (1, 2), (63, 3), (64, 4), (65, 5), (4097, 6), (20000, 7)])
def test_acceleration_structure_builder(exe, T, seed):
    run(exe, "accel", T, seed)


def test_eta_table(exe):
    run(exe, "eta")


def test_oracle_under_asan():
    """oracle/hrt_oracle.c with ASan + UBSan, from Python (runtime preloaded), on the C1 fixture."""
    p = subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "asan"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan.so to preload")
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import os; os.environ['HRT_ORACLE_LIB'] = os.path.join(%r, 'oracle', 'liboracle_asan.so')\n"
        "import numpy as np\n"
        "from oracle import oracle\n"
        "from tests import configs as K\n"
        "for c in (K.small(K.C1, 2000), K.small(K.C4_DOPPLER, 300), K.small(K.C5, 64), K.small(K.COINCIDENT, 500)):\n"
        "    r = oracle.compute_paths(*K.args(c), num_threads=2)\n"
        "    assert r['extras']['live'][0] > 0\n"
        "print('ORACLE_ASAN_OK')\n" % (REPO, REPO))
    env = dict(ENV, LD_PRELOAD=libasan + (":" + libubsan if os.path.exists(libubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:exitcode=99")
    q = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert q.returncode == 0 and "ORACLE_ASAN_OK" in q.stdout, q.stdout[-1000:] + q.stderr[-3000:]


@pytest.mark.parametrize("target", ["parallel_tsan", "parallel_asan"])
def test_parallel_for_of_the_host_writers(target):
    """csrc/host/parallel.c (helper threads parked between loops, one pool per calling thread) under
    ThreadSanitizer and under ASan + UBSan: many loops of changing size and thread count, release and
    restart, four calling threads at once; every index of every loop visited exactly once."""
    p = subprocess.run(["make", "-C", ASAN_DIR, target], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    env = dict(ENV, TSAN_OPTIONS="halt_on_error=1:exitcode=97")
    r = subprocess.run([os.path.join(ASAN_DIR, target)], env=env, capture_output=True, text=True, timeout=600)
    if target == "parallel_tsan" and r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow here (address-space layout of this kernel)")
    assert r.returncode == 0 and "PARALLEL_OK" in r.stdout, r.stdout[-500:] + r.stderr[-3000:]
    assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
