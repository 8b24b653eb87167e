"""The Sionna importer (hrt_scene_import_sionna + the hrt_import_sionna CLI)
against what the reference's importer produced for the same inputs
(tests/golden/sionna_fixture/scene_expected.hrt, recorded by make_sionna_fixture.py), and the
two built-in scenes against the bundled .hrt files -- byte for byte."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from hermespy_rt_amd import LIB_DIR, abi, lib
from oracle import oracle

from . import configs as K

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sionna_fixture")
CLI = os.path.join(LIB_DIR, "hrt_import_sionna")


def _import(path):
    L = lib.load()
    sc = abi.Scene()
    lib.check(L.hrt_scene_import_sionna(str(path).encode(), C.byref(sc)), "hrt_scene_import_sionna")
    return L, sc


def test_library_call_matches_reference_tool(tmp_path):
    L, sc = _import(os.path.join(FIX, "scene.xml"))
    out = tmp_path / "o.hrt"
    L.scene_save(C.byref(sc), str(out).encode())
    abi.free_scene(sc)
    assert out.read_bytes() == open(os.path.join(FIX, "scene_expected.hrt"), "rb").read()
    m = oracle.read_hrt(str(out))
    assert [x["material_index"] for x in m] == [1, 6, 0, 13]     # unknown material -> air
    assert [len(x["idx"]) for x in m] == [2, 12, 9, 4]


def test_cli_matches_reference_tool_and_exit_codes(tmp_path):
    subprocess.check_call([CLI, os.path.join(FIX, "scene.xml")], cwd=tmp_path)
    assert (tmp_path / "scene.hrt").read_bytes() == open(os.path.join(FIX, "scene_expected.hrt"), "rb").read()
    subprocess.check_call([CLI, os.path.join(FIX, "scene.xml"), str(tmp_path / "named.hrt")])
    assert (tmp_path / "named.hrt").exists()
    assert subprocess.run([CLI, "/nonexistent/scene.xml"], capture_output=True).returncode == 8
    assert subprocess.run([CLI], capture_output=True).returncode == 1


@pytest.mark.parametrize("name", ["box", "simple_reflector"])
def test_builtin_scenes_are_the_bundled_files(name, tmp_path):
    out = tmp_path / "b.hrt"
    subprocess.check_call([CLI, "/anywhere/%s.xml" % name, str(out)])
    assert out.read_bytes() == open(os.path.join(K.SC, name + ".hrt"), "rb").read()


def test_csv_overrides_and_missing_csv(tmp_path):
    d = tmp_path / "s"
    shutil.copytree(FIX, d)
    os.remove(d / "scene.csv")                                   # optional here
    L, sc = _import(d / "scene.xml")
    got = abi.scene_to_numpy(sc)
    abi.free_scene(sc)
    assert [m["material_index"] for m in got] == [1, 6, 0, 13]
    shutil.copy(d / "scene_overrides.csv", d / "scene.csv")
    L, sc = _import(d / "scene.xml")
    got = abi.scene_to_numpy(sc)
    abi.free_scene(sc)
    assert [m["material_index"] for m in got] == [1, 13, 0, 4]
    assert np.array_equal(got[1]["velocity"], np.array([1.5, -2, 0.25], np.float32))
    assert np.array_equal(got[3]["velocity"], np.array([0, 0, 9], np.float32))
    assert np.array_equal(got[0]["velocity"], np.zeros(3, np.float32))


def test_errors_are_status_codes(tmp_path):
    L = lib.load()
    sc = abi.Scene()
    assert L.hrt_scene_import_sionna(b"/nonexistent/x.xml", C.byref(sc)) != 0
    assert b"cannot open" in L.hrt_last_error()
    p = tmp_path / "empty.xml"
    p.write_text("<scene></scene>")
    assert L.hrt_scene_import_sionna(str(p).encode(), C.byref(sc)) != 0
    p2 = tmp_path / "bad.txt"
    p2.write_text("x")
    assert L.hrt_scene_import_sionna(str(p2).encode(), C.byref(sc)) != 0


def test_imported_scene_traces_like_any_other(tmp_path):
    """an imported scene is a normal .hrt for the oracle (and, on a GPU, the product)"""
    L, sc = _import(os.path.join(FIX, "scene.xml"))
    out = tmp_path / "o.hrt"
    L.scene_save(C.byref(sc), str(out).encode())
    abi.free_scene(sc)
    r = oracle.compute_paths(str(out), [[3, 2, 5]], [[-4, 1, 6]], [[0, 0, 0]], [[0, 0, 0]], 3.5, 2000, 2)
    assert int(r["extras"]["live"][1]) > 0
