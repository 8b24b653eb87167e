"""The C-ABI shared library: loads without a GPU, exports exactly what include/*.h declare,
struct layouts match the reference's, host-only entry points work, and compute entry points
fail loudly (no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from hermespy_rt_amd import abi, lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(REPO, "include")


def _declared_functions():
    names = set()
    for h in ("hermespy_rt.h", "hrt_device.h"):
        src = open(os.path.join(INC, h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"static inline[^{]*\{.*?\n\}", "", src, flags=re.S)
        for m in re.finditer(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", src, flags=re.M):
            names.add(m.group(1))
    return names


def test_exports_match_headers(product_lib):
    declared = _declared_functions()
    assert declared == set(lib.EXPORTED), declared ^ set(lib.EXPORTED)
    for n in declared:
        assert hasattr(product_lib, n), n
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert exported == declared, exported ^ declared


def test_struct_layout_matches_c(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hrt_device.h"\n'
                    'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(Vec3), sizeof(Ray),'
                    'sizeof(Mesh), sizeof(Scene), sizeof(ChannelInfo), sizeof(RaysInfo), sizeof(hrt_shard),'
                    'sizeof(hrt_layout), sizeof(hrt_kernel_times), sizeof(hrt_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", INC, str(prog), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    want = [C.sizeof(t) for t in (abi.Vec3, abi.Ray, abi.Mesh, abi.Scene, abi.ChannelInfo, abi.RaysInfo,
                                  lib.Shard, lib.Layout, lib.KernelTimes, lib.Stats)]
    assert got == want
    assert got[:6] == [12, 24, 56, 16, 72, 24]   # the reference's struct sizes (x86-64)


def test_headers_compile_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "hermespy_rt.h"\n#include "hrt_device.h"\nint main(void){return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", INC, "-c", str(src), "-o", str(tmp_path / "t.o")])
    subprocess.check_call(["g++", "-x", "c++", "-Wall", "-Werror", "-I", INC, "-c", str(src), "-o", str(tmp_path / "t2.o")])


def test_version_and_error_strings(product_lib):
    assert b"gfx950" in product_lib.hrt_version()
    assert isinstance(product_lib.hrt_last_error(), bytes)


def _have_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_have_gpu(), reason="checks the no-device behaviour")
def test_compute_fails_loudly_without_device(product_lib):
    """No HIP device -> HRT_E_HIP from the _ex entry point, and exit(70) from the drop-in
    compute_paths(); never a silent CPU result."""
    from . import configs as K
    c = K.small(K.C1, 64)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from hermespy_rt_amd import abi, lib\nfrom tests import configs as K\n"
            "abi.run_compute_paths(lib.load(), *K.args(K.small(K.C1, 64)))\n" % REPO)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode == 70, (p.returncode, p.stderr[-500:])
    assert "compute_paths failed" in p.stderr
    # status-returning variant
    scene = product_lib.scene_load(c["scene_path"].encode())
    h = C.c_void_p()
    v = np.zeros((1, 3), np.float32)
    V3 = C.POINTER(abi.Vec3)
    rc = product_lib.hrt_problem_create(C.byref(scene), v.ctypes.data_as(V3), v.ctypes.data_as(V3),
                                        v.ctypes.data_as(V3), v.ctypes.data_as(V3), C.c_float(3.0), 1, 1, 0, C.byref(h))
    abi.free_scene(scene)
    assert rc == -3 and b"HIP" in product_lib.hrt_last_error()


def test_tracer_requires_device_or_works():
    from hermespy_rt_amd.device import Tracer
    from . import configs as K
    c = K.small(K.C1, 64)
    if _have_gpu():
        pytest.skip("GPU present")
    with pytest.raises(lib.HrtError):
        Tracer(c["scene_path"], c["rx_pos"], c["tx_pos"], c["rx_vel"], c["tx_vel"], 3.0, 64, 1)


def test_product_never_touches_the_oracle():
    """Policy: nothing under hermespy-rt_amd/ (nor the root shim) imports, links or opens
    anything under oracle/."""
    bad = []
    root = os.path.join(REPO, "hermespy-rt_amd")
    for dp, dn, fn in os.walk(root):
        if "build" in dp or "/lib" in dp:
            continue
        for f in fn:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp", ".map")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for m in re.finditer(r"oracle", txt):
                    line = txt[txt.rfind("\n", 0, m.start()) + 1: txt.find("\n", m.end())]
                    if not re.search(r"^\s*(#|//|\*|/\*)|oracle/libm_probe|pinned|Pinned|probe", line):
                        bad.append((f, line.strip()))
    assert not bad, bad
    deps = subprocess.check_output(["ldd", lib.LIB_PATH], text=True)
    assert "oracle" not in deps


REF_INC = "/root/reference/inc"

CALLER_SRC = r'''
/* a caller written against the REFERENCE's headers only (INTEGRATION.md section 1) */
#include "compute_paths.h"
#include "scene.h"
#include <stdio.h>
#include <string.h>
int main(int argc, char **argv)
{
    Scene s = scene_load(argv[1]);
    if (s.num_meshes != 1 || s.meshes[0].num_triangles != 12 || s.meshes[0].ns != NULL) return 2;
    scene_save(&s, argv[2]);
    /* the entry point resolves against the product library (calling it needs a GPU) */
    void (*fp)(Scene *, Vec3 *, Vec3 *, Vec3 *, Vec3 *, float, size_t, size_t, size_t, size_t, ChannelInfo *,
               RaysInfo *, ChannelInfo *, RaysInfo *) = compute_paths;
    if (argc > 3) {   /* GPU box: one tiny call, box.hrt, the SURVEY 8c anchor inputs */
        Vec3 rx = {2, 1, 1.5f}, tx = {0, 0, 2.5f}, z = {0, 0, 0};
        enum { NP = 1000 };
        static Vec3 dl_rx[1], dl_tx[1], ds_rx[NP], ds_tx[NP];
        static float l[6][1], sc[6][NP];
        ChannelInfo los = {1, dl_rx, dl_tx, l[0], l[1], l[2], l[3], l[4], l[5]};
        ChannelInfo scat = {NP, ds_rx, ds_tx, sc[0], sc[1], sc[2], sc[3], sc[4], sc[5]};
        static Ray lr[1], sr[2 * NP];
        static uint8_t la[8], sa[2 * (NP / 8 + 1)];
        RaysInfo rl = {1, 1, lr, la}, rs = {1, NP, sr, sa};
        fp(&s, &rx, &tx, &z, &z, 3.0f, 1, 1, NP, 1, &los, &rl, &scat, &rs);
        printf("los_tau %.9g a %.9g\n", l[4][0], l[0][0]);
        if (!s.meshes[0].ns) return 3;
    }
    free_scene(&s);
    printf("caller ok %p\n", (void *)fp);
    return 0;
}
'''


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="needs the reference headers (build container only)")
def test_caller_built_against_reference_headers_links_unchanged(tmp_path):
    """INTEGRATION.md section 1, literally: a C caller compiled with -I/root/reference/inc (not our
    headers) links against libhermespy_rt_amd.so and runs: scene_load returns the reference's Scene
    by value, scene_save writes the same bytes, compute_paths resolves (and, with a GPU, is called
    and returns the SURVEY 8c LoS anchor)."""
    src = tmp_path / "caller.c"
    src.write_text(CALLER_SRC)
    exe = tmp_path / "caller"
    libdir = os.path.dirname(lib.LIB_PATH)
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", REF_INC, str(src), "-o", str(exe), "-L", libdir,
                           "-lhermespy_rt_amd", "-Wl,-rpath," + libdir, "-lm"])
    box = os.path.join(REPO, "scenes", "box.hrt")
    out = tmp_path / "again.hrt"
    args = [str(exe), box, str(out)] + (["gpu"] if _have_gpu() else [])
    p = subprocess.run(args, capture_output=True, text=True)
    assert p.returncode == 0 and "caller ok" in p.stdout, (p.returncode, p.stdout, p.stderr[-500:])
    assert open(out, "rb").read() == open(box, "rb").read()
    if _have_gpu():
        assert "los_tau 8.17061885e-09 a 0.00324648875" in p.stdout, p.stdout
