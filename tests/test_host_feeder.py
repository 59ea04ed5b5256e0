"""Host side of the product (scene ingest, transforms, octree, C ABI surface) against the oracle. No GPU needed."""
import ctypes as C
import os
import re

import numpy as np
import pytest
from PIL import Image

import raytracer_amd as rt
from oracle_api import OracleScene
from scenes_common import ALL_SCENES, REFERENCE

ROOT = rt.ROOT


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name", ALL_SCENES)
def test_feeder_is_bit_identical_to_oracle(name):
    # odd, non-square size so that the frustum arithmetic (right = W*top/H) is exercised
    o = OracleScene("scenes/%s.json" % name, 640, 360, 3)
    s = rt.Scene("scenes/%s.json" % name, 640, 360, 3)
    assert np.array_equal(_bits(o.camera()), _bits(s.camera()))
    ov, on, om = o.tris()
    sv, sn, sm = s.triangles()
    assert np.array_equal(_bits(ov), _bits(sv))
    assert np.array_equal(_bits(on), _bits(sn))
    assert np.array_equal(om, sm)
    oi, si = o.info(), s.info()
    assert (oi["n_nodes"], oi["n_leaves"], oi["n_empty_leaves"], oi["n_refs"], oi["max_depth"], oi["max_leaf"]) == \
           (si["n_nodes"], si["n_leaves"], si["n_empty_leaves"], si["n_leaf_refs"], si["tree_depth"], si["max_leaf_size"])
    if oi["n_nodes"]:
        ok, ob, orf = o.tree()
        sk, sb, srf = s.tree()
        # depth-first preorder: same topology, same box corners, same leaf contents in the same order
        assert np.array_equal(ok, sk)
        assert np.array_equal(_bits(ob), _bits(sb))
        assert np.array_equal(orf, srf)


def test_dragon_tree_statistics_match_reference_run():
    exp = REFERENCE["trees"]["meshDragon"]
    i = rt.Scene("scenes/meshDragon.json", 64, 64, 3).info()
    assert i["n_triangles"] == exp["n_tris"] and i["n_nodes"] == exp["n_nodes"] and i["n_leaves"] == exp["n_leaves"]
    assert i["n_empty_leaves"] == exp["n_empty_leaves"] and i["n_leaf_refs"] == exp["n_refs"]
    assert i["tree_depth"] == exp["max_depth"] and i["max_leaf_size"] == exp["max_leaf"]
    assert i["has_bounce_materials"] == 0 and i["shadows_enabled"] == 1


def test_bounce_flag():
    assert rt.Scene("scenes/meshTransparentBunny.json", 64, 64, 3).info()["has_bounce_materials"] == 1
    assert rt.Scene("scenes/meshMedBunny.json", 64, 64, 3).info()["has_bounce_materials"] == 0


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    declared = set(re.findall(r"^(?:int|void\*?|const char\*)\s+(rt_[a-z_]+)\s*\(", header, re.M))
    assert declared == set(rt.ABI_SYMBOLS), declared ^ set(rt.ABI_SYMBOLS)
    L = C.CDLL(rt.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert "gfx950" in rt.version()


def test_error_codes(tmp_path):
    with pytest.raises(rt.RtError) as e:
        rt.Scene("scenes/does_not_exist.json", 64, 64, 3)
    assert e.value.status == -2
    with pytest.raises(rt.RtError) as e:   # the ray stack holds 10 rays: depth is limited to 9
        rt.Scene("scenes/sphere.json", 64, 64, 10)
    assert e.value.status == -4
    with pytest.raises(rt.RtError) as e:
        rt.Scene("scenes/sphere.json", 0, 64, 3)
    assert e.value.status == -1
    for w, h in ((65536, 4), (4, 65536), (-3, 4)):   # image sides are unsigned shorts in the reference (1..65535)
        with pytest.raises(rt.RtError) as e:
            rt.Scene("scenes/sphere.json", w, h, 3)
        assert e.value.status == -1
    assert rt.Scene("scenes/sphere.json", 65535, 1, 3).info()["n_spheres"] == 1
    bad = tmp_path / "bad.json"
    bad.write_text('{"camera": {"fov": 45.0, "position": [0,0,5], "up": [0,1,0], "target": [0,0,0]},'
                   ' "objects": [{"__type__": "sphere", "radius": 1, "position": [0,0,0], "materials": ["no_such"]}]}')
    with pytest.raises(rt.RtError) as e:
        rt.Scene(str(bad), 64, 64, 3)
    assert e.value.status == -3
    broken = tmp_path / "broken.json"
    broken.write_text('{"camera": {"fov": 45.0, ')
    with pytest.raises(rt.RtError) as e:
        rt.Scene(str(broken), 64, 64, 3)
    assert e.value.status == -3
    # the 8 meshes of this scene need bunny_high.obj, which the reference repository does not ship
    with pytest.raises(rt.RtError) as e:
        rt.Scene("scenes/meshHighBunny.json", 64, 64, 3)
    assert e.value.status == -2


def test_device_prepare_without_a_device_says_so():
    # (no CPU fallback: on a machine without a HIP device the optional bring-up call reports RT_ERR_DEVICE like the upload does)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    assert rt.lib().rt_device_prepare(0) == -5
    assert b"no HIP device" in rt.lib().rt_last_error()


def test_png_writer_roundtrip(tmp_path):
    rng = np.random.default_rng(7)
    for (h, w) in ((1, 1), (37, 53), (300, 257)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / ("t%dx%d.png" % (w, h)))
        rt.write_png(p, img)
        back = np.array(Image.open(p).convert("RGB"))
        assert np.array_equal(back, img)


def test_tree_does_not_depend_on_the_number_of_build_threads(monkeypatch):
    trees = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("RT_AMD_BUILD_THREADS", threads)
        s = rt.Scene("scenes/meshDragon.json", 64, 64, 3)
        k, b, r = s.tree()
        trees.append((k.copy(), b.copy(), r.copy()))
    for k, b, r in trees[1:]:
        assert np.array_equal(k, trees[0][0]) and np.array_equal(_bits(b), _bits(trees[0][1])) and np.array_equal(r, trees[0][2])


def test_version_names_the_tunables_in_force():
    # RT_AMD_* variables move thresholds and switch parts of the kernels off (measurements); a process that runs with any of
    # them says so in its version string, which every bench line and `run --stats` line carries
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); import raytracer_amd as rt; print(rt.version())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in [k for k in env if k.startswith("RT_AMD_")]:
        del env[k]
    plain = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60).stdout
    assert "non-default" not in plain and plain.startswith("rt_amd ")
    env["RT_AMD_NO_CULL"] = "1"
    env["RT_AMD_CULL_MIN_NODES"] = "64"
    knobs = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60).stdout
    assert "non-default" in knobs and "RT_AMD_NO_CULL=1" in knobs and "RT_AMD_CULL_MIN_NODES=64" in knobs
