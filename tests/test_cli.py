"""The `run` command line keeps the reference's contract (parsing/parsing.cu:145-175, main.cu:34-35,68-69)."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import raytracer_amd as rt

RUN = os.path.join(rt.ROOT, "raytracer_amd", "run")


def test_usage_without_scene_exits_zero():
    out = subprocess.run([RUN], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0
    assert out.stdout.startswith("Usage: ./raytracer -w W -h H -s scene.json")


def test_missing_mesh_reports_like_the_reference(tmp_path):
    out = subprocess.run([RUN, "-s", "scenes/meshHighBunny.json", "-i", str(tmp_path / "x.png")], cwd=rt.ASSETS,
                         stdout=subprocess.PIPE, text=True)
    assert out.returncode == 1
    assert "No meshes/bunny_high.obj has been found" in out.stdout
    assert "+ Width: 512" in out.stdout and "+ MaxRayDepth: 3" in out.stdout


@pytest.mark.gpu
def test_cli_renders_the_reference_image(tmp_path):
    img = str(tmp_path / "out.png")
    out = subprocess.run([RUN, "-s", "scenes/sphere.json", "-i", img, "--stats"], cwd=rt.ASSETS, stdout=subprocess.PIPE,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    assert "Using following defaults:" in out.stdout and "Oct taken seconds:" in out.stdout and "Rendering seconds:" in out.stdout
    ref = np.array(Image.open(os.path.join(rt.ROOT, "tests", "golden", "sphere_512_reference_image.png")).convert("RGB"))
    assert np.array_equal(np.array(Image.open(img).convert("RGB")), ref)
