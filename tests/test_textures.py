"""Texture materials (`brdf_color_texture_material` of resources.json).  The reference declares them and ships the images,
but its loader is an unfinished stub and its shading has no case for them (material/material.cu:263-301): nothing here
is pinned by reference behaviour ("parity unpinned").  What is tested: the library's image decoders against another
decoder (Pillow), the host feeder, and the HIP kernels against the CPU restatement of the same definition."""
import os

import numpy as np
import pytest

import raytracer_amd as rt
from oracle_api import ASSETS, OracleScene, provide_texture

TEXTURES = ["chess.png", "red.png", "bricks.jpg", "bricks_SPEC.jpg", "earthmap.jpg"]

TEXTURED_SCENE = """{
  "camera": {"fov": 50.0, "position": [0, 6, 14], "up": [0, 1, 0], "target": [0, 1.5, 0]},
  "objects": [
    {"__type__": "mesh", "file_path": "meshes/large_plane.obj", "materials": ["chess_lambert_bilinear"]},
    {"__type__": "mesh", "file_path": "meshes/sphere.obj", "materials": ["%(mesh_a)s", "white_bph_100"],
     "scaling": [2, 2, 2], "rotation": [10, 20, 30], "translation": [-3, 2, 0]},
    {"__type__": "sphere", "position": [2.5, 1.5, 1], "radius": 1.5, "materials": ["%(sphere_a)s", "%(sphere_b)s"]},
    {"__type__": "sphere", "position": [0, 0.8, 4], "radius": 0.8, "materials": ["red_lambert_bilinear"]}
  ],
  "lights": [
    {"__type__": "point_light", "position": [5, 8, 5], "color": [0.6, 0.6, 0.6]},
    {"__type__": "directional_light", "direction": [0.2, -1, -0.3], "color": [0.3, 0.3, 0.3]},
    {"__type__": "ambient_light", "color": [0.15, 0.15, 0.15]}
  ],
  "params": {"background_color": [0.1, 0.2, 0.3]}
}"""
PNG_ONLY = {"mesh_a": "chess_lambert_bilinear", "sphere_a": "red_lambert_nn", "sphere_b": "white_bph_1000"}
WITH_JPEG = {"mesh_a": "earth_lambert_bilinear", "sphere_a": "earth_lambert_nn", "sphere_b": "bricks_blinn_phong_bilinear"}


def _pillow(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(ASSETS, "textures", name)).convert("RGB"))


@pytest.mark.parametrize("name", TEXTURES)
def test_decoders_against_pillow(name):
    got, want = rt.load_image(os.path.join(ASSETS, "textures", name)), _pillow(name)
    assert got.shape == want.shape
    diff = np.abs(got.astype(int) - want.astype(int))
    if name.endswith(".png"):
        assert diff.max() == 0                     # lossless: every byte
    else:
        # baseline JPEG: this decoder rounds the exact inverse transform once, libjpeg uses an integer transform
        assert diff.max() <= 4 and diff.mean() <= 0.05 and (diff > 1).mean() <= 0.01, (diff.max(), diff.mean(), (diff > 1).mean())


def test_decoder_errors(tmp_path):
    with pytest.raises(rt.RtError) as e:
        rt.load_image(str(tmp_path / "missing.png"))
    assert e.value.status == -2
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not an image at all")
    with pytest.raises(rt.RtError) as e:
        rt.load_image(str(bad))
    assert e.value.status == -3
    cut = tmp_path / "cut.jpg"
    cut.write_bytes(open(os.path.join(ASSETS, "textures", "bricks.jpg"), "rb").read()[:300])
    with pytest.raises(rt.RtError):
        rt.load_image(str(cut))


def test_host_feeder_loads_textured_scenes(tmp_path):
    p = tmp_path / "tex.json"
    p.write_text(TEXTURED_SCENE % WITH_JPEG)
    sc = rt.Scene(str(p), 64, 48, 3, base_dir=ASSETS)
    assert sc.info()["n_triangles"] > 0
    # a texture whose file is missing is an I/O error, as for a missing mesh
    res = tmp_path / "resources.json"
    res.write_text(open(os.path.join(ASSETS, "resources.json")).read().replace("textures/chess.png", "textures/nowhere.png"))
    with pytest.raises(rt.RtError) as e:
        rt.Scene(str(p), 64, 48, 3, base_dir=ASSETS, resources=str(res))
    assert e.value.status == -2 and "nowhere.png" in str(e.value)
    # scenes that use no texture material never touch the image files
    rt.Scene("scenes/sphere.json", 32, 32, 3, resources=str(res))


@pytest.mark.gpu
@pytest.mark.parametrize("which,w,h", [("png", 256, 256), ("png", 203, 117), ("jpeg", 256, 256)])
def test_textured_scene_parity(tmp_path, which, w, h):
    from test_gpu_parity import _check_scene
    # The checker is handed decoded pixels: Pillow's for the PNG textures (another decoder than the library's); for the JPEG
    # ones the library's own (Pillow's differ by a level here and there, see test_decoders_against_pillow), so that this
    # test is about the lookup and the shading, not about inverse-transform rounding.
    for name in TEXTURES:
        provide_texture("textures/" + name, _pillow(name) if name.endswith(".png") else rt.load_image(os.path.join(ASSETS, "textures", name)))
    p = tmp_path / ("tex_%s.json" % which)
    p.write_text(TEXTURED_SCENE % (PNG_ONLY if which == "png" else WITH_JPEG))
    frac_ok, n_diff, max_diff = _check_scene(str(p), w, h, base_dir=ASSETS)
    print("textured scene (%s) %dx%d: %d pixels differ from the checker, largest difference %d" % (which, w, h, n_diff, max_diff))
    # the textures are seen: the frame differs from the same scene with plain materials
    ds = rt.DeviceScene(rt.Scene(str(p), w, h, 3, base_dir=ASSETS))
    rgb, _, _, _ = ds.render_host(want_ids=False)
    q = tmp_path / "plain.json"
    q.write_text(TEXTURED_SCENE % {"mesh_a": "white_lambert", "sphere_a": "white_lambert", "sphere_b": "white_bph_1000"})
    plain, _, _, _ = rt.DeviceScene(rt.Scene(str(q), w, h, 3, base_dir=ASSETS)).render_host(want_ids=False)
    assert (np.abs(rgb.astype(int) - plain.astype(int)).max(axis=2) > 8).mean() > 0.02
