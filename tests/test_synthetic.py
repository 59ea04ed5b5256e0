"""Synthetic benchmark mesh: generator properties and host-side ingest against the oracle. No GPU needed."""
import hashlib

import numpy as np
import pytest

import raytracer_amd as rt
from oracle_api import OracleScene
from raytracer_amd.synthetic import icosphere, write_icosphere_scene


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_icosphere_is_a_closed_outward_oriented_unit_mesh(k):
    v, f = icosphere(k)
    assert len(f) == 20 * 4 ** k and len(v) == 10 * 4 ** k + 2          # Euler: V - E + F = 2 with E = 3F/2
    assert np.abs(np.linalg.norm(v, axis=1) - 1.0).max() < 1e-12
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    directed = set(map(tuple, e.tolist()))
    assert len(directed) == len(e)                                        # no directed edge twice
    assert all((b, a) in directed for a, b in directed)                   # every edge shared by exactly two faces
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    assert ((n * v[f].mean(1)).sum(1) > 0).all()                          # counter-clockwise seen from outside


def test_generator_is_deterministic(tmp_path):
    a, _ = write_icosphere_scene(str(tmp_path / "a"), 3)
    b, _ = write_icosphere_scene(str(tmp_path / "b"), 3)
    da = hashlib.sha256(open(str(tmp_path / "a" / "meshes" / "icosphere_k3.obj"), "rb").read()).hexdigest()
    db = hashlib.sha256(open(str(tmp_path / "b" / "meshes" / "icosphere_k3.obj"), "rb").read()).hexdigest()
    assert da == db and open(a).read() == open(b).read()


def test_host_ingest_of_the_synthetic_scene_matches_the_oracle(tmp_path):
    path, n = write_icosphere_scene(str(tmp_path), 4)
    o = OracleScene(path, 320, 180, 3, base_dir=str(tmp_path))
    s = rt.Scene(path, 320, 180, 3, base_dir=str(tmp_path))
    assert s.info()["n_triangles"] == n + 2                               # sphere + the two triangles of the plane
    ov, on, om = o.tris()
    sv, sn, sm = s.triangles()
    assert np.array_equal(ov.view(np.uint32), sv.view(np.uint32)) and np.array_equal(on.view(np.uint32), sn.view(np.uint32))
    ok, ob, orf = o.tree()
    sk, sb, srf = s.tree()
    assert np.array_equal(ok, sk) and np.array_equal(ob.view(np.uint32), sb.view(np.uint32)) and np.array_equal(orf, srf)
    # vertex normals come from the file: position / radius about the centre
    c = np.array([0.0, 6.0, 0.0], np.float32)
    sph = sv.reshape(-1, 3, 3)[:n]
    nn = sn.reshape(-1, 3, 3)[:n]
    assert np.abs((sph - c) / 6.0 - nn).max() < 1e-6
