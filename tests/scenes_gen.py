"""Synthetic .hrt scenes for the code paths the four bundled scenes do not reach: more than
64 / 256 / 1024 triangles (several culling rounds, the unculled tail), tables too big for LDS,
moving meshes (non-zero Doppler terms), all 17 materials, exact ties (duplicated triangles),
degenerate triangles, shared edges.  Deterministic (seeded)."""
import os
import struct

import numpy as np


def write_hrt(path, meshes):
    """meshes: list of dict(vs [nv,3] f32, idx [nt,3] u32, material_index, velocity[3])."""
    with open(path, "wb") as f:
        f.write(b"HRT")
        f.write(struct.pack("<I", len(meshes)))
        for m in meshes:
            vs = np.ascontiguousarray(m["vs"], np.float32)
            idx = np.ascontiguousarray(m["idx"], np.uint32)
            f.write(struct.pack("<I", len(vs)))
            f.write(vs.tobytes())
            f.write(struct.pack("<I", len(idx)))
            f.write(idx.tobytes())
            f.write(struct.pack("<I", int(m["material_index"])))
            f.write(np.asarray(m["velocity"], np.float32).tobytes())


def _box(center, size):
    c, s = np.asarray(center, np.float32), np.asarray(size, np.float32) / 2
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * s + c
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1],
                  [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.uint32)
    return v, f


def room_with_clutter(path, n_boxes, seed=1, moving=True, tilt=False, scale=1.0, max_meshes=900):
    """A 40 x 30 x 12 m room (inward walls = one box mesh) with n_boxes random boxes inside:
    12 * (n_boxes + 1) triangles.  Materials cycle through all 17; odd meshes move.
    `scale` stretches the room (not the boxes): scale = (n_boxes / 500) ** (1/3) keeps the clutter
    density of the 500-box room.  The file format allows 1000 meshes (src/scene.c:54 of the
    reference): beyond max_meshes boxes, consecutive boxes share a mesh."""
    rng = np.random.default_rng(seed)
    meshes = []
    v, f = _box([0, 0, 6 * scale], [40 * scale, 30 * scale, 12 * scale])
    meshes.append(dict(vs=v, idx=f, material_index=1, velocity=[0, 0, 0]))
    per = max(1, -(-n_boxes // max_meshes))
    cur = None
    for i in range(n_boxes):
        c = rng.uniform([-18 * scale, -13 * scale, 0.5], [18 * scale, 13 * scale, 12 * scale - 2])
        sz = rng.uniform(0.3, 2.5, 3)
        v, f = _box(c, sz)
        if tilt:   # rotate about z and x so normals are not axis aligned
            a, b = rng.uniform(0, np.pi, 2)
            Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
            v = ((v - c) @ (Rz @ Rx).T + c).astype(np.float32)
        if i % per == 0:
            k = i // per
            vel = rng.uniform(-30, 30, 3) if (moving and k % 2) else np.zeros(3)
            cur = dict(vs=v, idx=f, material_index=k % 17, velocity=vel)
            meshes.append(cur)
        else:
            cur["idx"] = np.concatenate([cur["idx"], f + len(cur["vs"])]).astype(np.uint32)
            cur["vs"] = np.concatenate([cur["vs"], v]).astype(np.float32)
    write_hrt(path, meshes)
    return 12 * (n_boxes + 1)


def city(path, n_side, seed=3, pitch=30.0, max_meshes=900, moving=False):
    """A Sionna-like city block grid: n_side x n_side buildings (boxes without a bottom face: 10
    triangles each; footprints 12-24 m, heights 8-60 m, each turned a few degrees about z) on a
    `pitch` metre grid, over a two-triangle ground plane.  10 * n_side^2 + 2 triangles; consecutive
    buildings share a mesh beyond max_meshes.  Returns (triangle count, half extent)."""
    rng = np.random.default_rng(seed)
    half = 0.5 * n_side * pitch
    g = np.array([[-half - 50, -half - 50, 0], [half + 50, -half - 50, 0], [half + 50, half + 50, 0],
                  [-half - 50, half + 50, 0]], np.float32)
    meshes = [dict(vs=g, idx=np.array([[0, 1, 2], [0, 2, 3]], np.uint32), material_index=1, velocity=[0, 0, 0])]
    n = n_side * n_side
    per = max(1, -(-n // max_meshes))
    cur = None
    k = 0
    for iy in range(n_side):
        for ix in range(n_side):
            cx, cy = (ix + 0.5) * pitch - half, (iy + 0.5) * pitch - half
            w, dpt, h = rng.uniform(12, 24), rng.uniform(12, 24), rng.uniform(8, 60)
            v, f = _box([cx, cy, h / 2], [w, dpt, h])
            a = rng.uniform(-0.2, 0.2)
            Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            c = np.array([cx, cy, h / 2])
            v = ((v - c) @ Rz.T + c).astype(np.float32)
            # drop the bottom face (z = 0): the two triangles whose vertices all have the lowest z
            zmin = v[:, 2].min()
            f = np.array([t for t in f if not np.all(np.isclose(v[t, 2], zmin))], np.uint32)
            if k % per == 0:
                vel = rng.uniform(-10, 10, 3) if (moving and (k // per) % 5 == 0) else np.zeros(3)
                cur = dict(vs=v, idx=f, material_index=(k // per) % 17, velocity=vel)
                meshes.append(cur)
            else:
                cur["idx"] = np.concatenate([cur["idx"], f + len(cur["vs"])]).astype(np.uint32)
                cur["vs"] = np.concatenate([cur["vs"], v]).astype(np.float32)
            k += 1
    write_hrt(path, meshes)
    return 10 * n + 2, half


def nasty(path):
    """Ties and degeneracies: a floor quad given twice (every floor hit is an exact tie between
    two different (mesh, face) pairs), two triangles sharing an edge hit exactly, a zero-area
    triangle (NaN normal), a sliver, a huge far triangle."""
    quad_v = np.array([[-20, -20, 0], [20, -20, 0], [20, 20, 0], [-20, 20, 0]], np.float32)
    quad_f = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    meshes = [
        dict(vs=quad_v, idx=quad_f, material_index=1, velocity=[0, 0, 0]),
        dict(vs=quad_v.copy(), idx=quad_f.copy(), material_index=13, velocity=[5, 0, 0]),   # duplicate
        dict(vs=np.array([[0, 0, 5], [3, 0, 5], [0, 3, 5], [3, 3, 5]], np.float32),
             idx=np.array([[0, 1, 2], [1, 3, 2]], np.uint32), material_index=4, velocity=[0, 0, 1]),
        dict(vs=np.array([[1, 1, 2], [1, 1, 2], [2, 2, 3]], np.float32),            # zero area
             idx=np.array([[0, 1, 2]], np.uint32), material_index=2, velocity=[0, 0, 0]),
        dict(vs=np.array([[-5, 0, 1], [5, 0, 1], [0, 1e-4, 1]], np.float32),         # sliver
             idx=np.array([[0, 1, 2]], np.uint32), material_index=6, velocity=[0, 0, 0]),
        dict(vs=np.array([[-1e3, -1e3, 50], [1e3, -1e3, 50], [0, 2e3, 50]], np.float32),
             idx=np.array([[0, 1, 2]], np.uint32), material_index=16, velocity=[0, 0, -3]),
    ]
    write_hrt(path, meshes)


def cfg(scene_path, rx, tx, np_, nb, f=3.5, rx_vel=None, tx_vel=None):
    z = [0.0, 0.0, 0.0]
    return dict(scene_path=str(scene_path), rx_pos=rx, tx_pos=tx, rx_vel=rx_vel or [z] * len(rx),
                tx_vel=tx_vel or [z] * len(tx), f_ghz=f, num_paths=np_, num_bounces=nb)
