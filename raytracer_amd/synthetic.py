"""Synthetic benchmark mesh: the unit icosahedron subdivided k times, over the reference's large plane.

Every subdivision splits a triangle into four through its edge midpoints, pushed back onto the unit sphere
(20 * 4**k triangles: k=6 -> 81 920, k=7 -> 327 680).  The sphere is written with radius 6, centred at
(0, 6, 0), with vertex normals position/radius, as a Wavefront OBJ of the subset the scene loader reads
(parsing/parsing.cu:21-130: one block of `v`, one of `vn`, then `f v//n v//n v//n`).  Camera, lights and
plane are those of scenes/meshDragon.json, the material list is ["red_lambert", "white_bph_1000"].
No random numbers: the same k always gives the same bytes.
"""
import json
import os
import shutil

import numpy as np

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def icosphere(k):
    """Unit-sphere vertices (float64, N x 3) and faces (int32, M x 3) after k subdivisions."""
    p = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([(-1, p, 0), (1, p, 0), (-1, -p, 0), (1, -p, 0), (0, -1, p), (0, 1, p), (0, -1, -p), (0, 1, -p),
                  (p, 0, -1), (p, 0, 1), (-p, 0, -1), (-p, 0, 1)], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
                  (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
                  (8, 6, 7), (9, 8, 1)], dtype=np.int64)
    for _ in range(k):
        n = len(v)
        # one midpoint per undirected edge, numbered in order of first appearance
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.minimum(e[:, 0], e[:, 1]) * n + np.maximum(e[:, 0], e[:, 1])
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        order = np.argsort(np.argsort(first))          # rank of each unique edge by first appearance
        mid_of = n + order[inv]
        a, b = uniq // n, uniq % n
        m = v[a] + v[b]
        m /= np.linalg.norm(m, axis=1, keepdims=True)
        mids = np.empty_like(m)
        mids[order] = m
        v = np.concatenate([v, mids])
        nf = len(f)
        ab, bc, ca = mid_of[:nf], mid_of[nf:2 * nf], mid_of[2 * nf:]
        f = np.concatenate([np.stack([f[:, 0], ab, ca], 1), np.stack([f[:, 1], bc, ab], 1),
                            np.stack([f[:, 2], ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return v, f.astype(np.int32)


def write_icosphere_scene(out_dir, k, radius=6.0, centre=(0.0, 6.0, 0.0)):
    """Writes meshes/icosphere_k<k>.obj, meshes/large_plane.obj and scenes/icosphere_k<k>.json under out_dir.
    Returns (scene_path, n_triangles); load it with base_dir=out_dir."""
    v, f = icosphere(k)
    os.makedirs(os.path.join(out_dir, "meshes"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "scenes"), exist_ok=True)
    pos = (v * radius + np.asarray(centre)).astype(np.float32)
    nrm = v.astype(np.float32)
    obj = os.path.join(out_dir, "meshes", "icosphere_k%d.obj" % k)
    with open(obj, "w") as o:
        o.write("".join("v %.9g %.9g %.9g\n" % tuple(p) for p in pos))
        o.write("".join("vn %.9g %.9g %.9g\n" % tuple(n) for n in nrm))
        o.write("".join("f %d//%d %d//%d %d//%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in f))
    shutil.copyfile(os.path.join(ASSETS, "meshes", "large_plane.obj"), os.path.join(out_dir, "meshes", "large_plane.obj"))
    with open(os.path.join(ASSETS, "scenes", "meshDragon.json")) as s:
        scene = json.load(s)
    scene["objects"][0] = {"__type__": "mesh", "file_path": "meshes/icosphere_k%d.obj" % k,
                           "materials": ["red_lambert", "white_bph_1000"]}
    path = os.path.join(out_dir, "scenes", "icosphere_k%d.json" % k)
    with open(path, "w") as s:
        json.dump(scene, s, indent=1)
    return path, len(f)
