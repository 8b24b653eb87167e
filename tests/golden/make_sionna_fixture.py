#!/usr/bin/env python3
"""Synthesise a small Sionna/Mitsuba-style scene (XML + binary PLY + CSV) and record what the
REFERENCE importer (oracle/_ref/sfs, built in place from src/scene_fromSionna.c) makes of it.

Inputs are written to tests/golden/sionna_fixture/ (they are ours: generated here, seeded);
the expected output scene_expected.hrt is the reference tool's.  The CSV has the header only:
that is the one CSV the reference tool accepts with fully defined behaviour.
"""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "sionna_fixture")


def write_ply(path, vs, faces, rng):
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\ncomment synthetic fixture\n")
        f.write(b"element vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % len(vs))
        f.write(b"property float s\nproperty float t\n")
        f.write(b"element face %d\nproperty list uchar int vertex_index\nend_header\n" % len(faces))
        for v in vs:
            f.write(struct.pack("<5f", *v, *rng.uniform(0, 1, 2)))
        for t in faces:
            f.write(struct.pack("<B3i", 3, *t))


def main():
    rng = np.random.default_rng(7)
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(os.path.join(OUT, "meshes"))
    shapes = [("ground", "concrete", 4, 2), ("tower-1", "glass2", 8, 12), ("blob", "unobtainium", 12, 9),
              ("roof", "metal", 5, 4)]
    xml = ['<scene version="2.1.0">', '  <default name="spp" value="4096"/>',
           '  <bsdf type="twosided" id="mat-itu_concrete"><bsdf type="diffuse"/></bsdf>']
    for name, mat, nv, nt in shapes:
        vs = rng.uniform(-20, 20, (nv, 3)).astype(np.float32)
        faces = np.stack([rng.choice(nv, 3, replace=False) for _ in range(nt)]).astype(np.int32)
        write_ply(os.path.join(OUT, "meshes", name + ".ply"), vs, faces, rng)
        xml += ['  <shape type="ply" id="mesh-%s" name="%s">' % (name, name),
                '    <string name="filename" value="meshes/%s.ply"/>' % name,
                '    <boolean name="face_normals" value="true"/>',
                '    <ref id="mat-itu_%s" name="bsdf"/>' % mat, '  </shape>']
    xml.append("</scene>\n")
    open(os.path.join(OUT, "scene.xml"), "w").write("\n".join(xml))
    open(os.path.join(OUT, "scene.csv"), "w").write("name,material_index,velocity_x,velocity_y,velocity_z\n")
    # a second CSV exercising overrides (behaviour defined by OUR importer only; see its header)
    open(os.path.join(OUT, "scene_overrides.csv"), "w").write(
        "name,material_index,velocity_x,velocity_y,velocity_z\ntower-1,13,1.5,-2,0.25\nroof,4,0,0,9\n")
    sfs = os.path.join(REPO, "oracle", "_ref", "sfs")
    if not os.path.exists(sfs):
        sys.exit("oracle/_ref/sfs missing: run `make -C oracle ref` where /root/reference is mounted")
    subprocess.check_call([sfs, os.path.join(OUT, "scene.xml")], cwd=OUT)
    os.replace(os.path.join(OUT, "scene.hrt"), os.path.join(OUT, "scene_expected.hrt"))
    print("fixture written:", sorted(os.listdir(OUT)), os.path.getsize(os.path.join(OUT, "scene_expected.hrt")), "bytes")


if __name__ == "__main__":
    main()
