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
