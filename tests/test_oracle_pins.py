"""The CPU restatement in oracle/ against everything the reference holds or recorded for this path."""
import hashlib
import os

import numpy as np
import pytest
from PIL import Image

from oracle_api import OracleScene
from scenes_common import COUNTED_SCENES, COUNTER_KEYS, HERE, REFERENCE


def test_sphere_scene_reproduces_reference_image_png():
    # image.png in the reference repository is scenes/sphere.json at the 512x512 defaults.
    ref = np.array(Image.open(os.path.join(HERE, "golden", "sphere_512_reference_image.png")).convert("RGB"))
    rgb, ids, ts, ct = OracleScene("scenes/sphere.json", 512, 512, 3).render()
    assert ref.shape == rgb.shape
    assert np.array_equal(ref, rgb)
    assert ct["rays"] == 512 * 512 and ct["sphere_tests"] == 512 * 512


@pytest.mark.parametrize("name", COUNTED_SCENES)
def test_work_counters_match_reference_run(name):
    exp = REFERENCE["scenes"][name]
    w, h = REFERENCE["size"]
    sc = OracleScene("scenes/%s.json" % name, w, h, REFERENCE["depth"])
    _, _, _, ct = sc.render(threads=4)
    for k in COUNTER_KEYS:
        assert ct[k] == exp[k], (name, k)
    assert ct["ub_leave_level0"] == exp.get("ub_leave_level0", 0)
    assert ct["ub_level_overflow"] == exp.get("ub_level_overflow", 0)
    assert ct["ub_stack_overflow"] == 0


@pytest.mark.parametrize("name", list(REFERENCE["trees"].keys()))
def test_octree_statistics_match_reference_run(name):
    exp = REFERENCE["trees"][name]
    info = OracleScene("scenes/%s.json" % name, 64, 64, 3).info()
    got = {"n_tris": info["n_tris"], "n_nodes": info["n_nodes"], "n_leaves": info["n_leaves"],
           "n_empty_leaves": info["n_empty_leaves"], "n_refs": info["n_refs"], "max_depth": info["max_depth"],
           "max_leaf": info["max_leaf"]}
    for k, v in exp.items():
        assert got[k] == v, (name, k)


def test_dragon_512_counters():
    exp = REFERENCE["meshDragon_512x512"]
    _, _, _, ct = OracleScene("scenes/meshDragon.json", 512, 512, 3).render(threads=8, want_ids=False)
    for k, v in exp.items():
        assert ct[k] == v, k


def test_dragon_4k_frame_digest_and_counters():
    # Headline configuration: 3840x2160, depth 3.  The digest is over the raw RGB8 frame.
    exp = REFERENCE["meshDragon_3840x2160"]
    rgb, _, _, ct = OracleScene("scenes/meshDragon.json", 3840, 2160, 3).render(threads=8, want_ids=False)
    for k in COUNTER_KEYS:
        assert ct[k] == exp[k], k
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == exp["rgb_sha256"]


def test_oracle_reproduces_the_committed_scene_fixtures():
    # tests/golden/scenes_128 (tools/make_goldens.py): the oracle of this checkout, built with this machine's compiler and
    # libm, against the arrays committed with the repository - ids, distance bits, RGB bytes and counters, all exact here
    # (the fixtures were made in this container image).
    import json
    import os
    import numpy as np
    from oracle_api import OracleScene
    from scenes_common import ALL_SCENES, HERE
    gdir = os.path.join(HERE, "golden", "scenes_128")
    with open(os.path.join(gdir, "manifest.json")) as f:
        man = json.load(f)
    assert sorted(man["scenes"]) == sorted(ALL_SCENES)
    for name, m in man["scenes"].items():
        g = np.load(os.path.join(gdir, name + ".npz"))
        rgb, ids, ts, ct = OracleScene("scenes/%s.json" % name, m["width"], m["height"], m["depth"]).render(threads=8)
        assert np.array_equal(ids, g["ids"]) and np.array_equal(ts.view(np.uint32), g["t_bits"]), name
        assert np.array_equal(rgb, g["rgb"]), (name, int((rgb != g["rgb"]).sum()))
        assert ct == m["counters"], (name, ct, m["counters"])
        import hashlib
        assert hashlib.sha256(np.ascontiguousarray(g["rgb"]).tobytes()).hexdigest() == m["sha256"]["rgb"]
