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


@pytest.mark.gpu
def test_cli_gpus_stats_and_dump_ids(tmp_path):
    # --gpus 1 goes through the band partition, the RCCL gather and the unpack kernel; it must write the image the
    # single-GPU path writes.  --stats is one JSON line with the time per phase; --dump-ids the primary-hit ids.
    import json
    a, b, ids = str(tmp_path / "a.png"), str(tmp_path / "b.png"), str(tmp_path / "ids.bin")
    base = [RUN, "-s", "scenes/meshMedBunny.json", "-w", "200", "-h", "150"]
    one = subprocess.run(base + ["-i", a, "--stats", "--dump-ids", ids], cwd=rt.ASSETS, stdout=subprocess.PIPE, text=True, timeout=120)
    assert one.returncode == 0, one.stdout
    many = subprocess.run(base + ["-i", b, "--gpus", "1", "--stats"], cwd=rt.ASSETS, stdout=subprocess.PIPE, text=True, timeout=120)
    assert many.returncode == 0, many.stdout
    assert np.array_equal(np.array(Image.open(a)), np.array(Image.open(b)))
    s1, s2 = json.loads(one.stdout.strip().splitlines()[-1]), json.loads(many.stdout.strip().splitlines()[-1])
    for key in ("parse_ms", "octree_ms", "upload_ms", "kernel_ms", "copy_ms", "png_ms", "mrays_per_s", "algorithmic_frac", "rays_walked"):
        assert key in s1 and key in s2
    assert s1["rays"] == s2["rays"] > 200 * 150 and s1["kernel_ms"] > 0 and s2["gpus"] == 1
    _, want_ids, _, _ = rt.DeviceScene(rt.Scene("scenes/meshMedBunny.json", 200, 150, 3)).render_host()
    assert np.array_equal(np.fromfile(ids, np.int32).reshape(150, 200), want_ids)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--gpus", "1"]])
def test_cli_frames_in_flight(tmp_path, extra):
    # --frames K: the pipeline set up once (rt_frame_ctx) renders the frame K times alone and K times with frames in flight;
    # every one of them must be the image that was written, and --stats carries both times
    import json
    out = subprocess.run([RUN, "-s", "scenes/meshMedBunny.json", "-w", "320", "-h", "200", "-i", str(tmp_path / "f.png"), "--stats", "--frames", "6"] + extra,
                         cwd=rt.ASSETS, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    st = json.loads(out.stdout.strip().splitlines()[-1])
    assert st["frames"] == 6 and st["frames_identical"] is True and st["frame_latency_ms"] > 0 and st["pipelined_ms_per_frame"] > 0
    assert st["frames_in_flight"] == 3


def test_cli_reports_an_unwritable_image(tmp_path):
    # without a GPU the upload fails first (status 1, no CPU fallback); the status is never 0 when no image was written
    out = subprocess.run([RUN, "-s", "scenes/sphere.json", "-i", str(tmp_path / "no_such_dir" / "x.png")], cwd=rt.ASSETS,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 1
