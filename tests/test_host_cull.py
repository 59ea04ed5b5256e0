"""The wave-level triangle skip of rt_math.h (rt_cull_record / rt_bundle_* / rt_cull_test) on the CPU: whenever it says that
no ray of a bundle can be accepted by a triangle, the literal restatement of triangle.cu:107-152 must reject every ray
of the bundle (tests/host_cull_check.cpp)."""
import os
import subprocess

import pytest

import raytracer_amd as rt

ROOT = rt.ROOT
EXE = os.path.join(ROOT, "tests", "host_cull_check")


@pytest.fixture(scope="module")
def harness():
    src = os.path.join(ROOT, "tests", "host_cull_check.cpp")
    libdir = os.path.join(ROOT, "raytracer_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + libdir, "-lrt_amd", "-Wl,-rpath," + libdir, "-o", EXE])
    return EXE


def _run(args):
    out = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    last = out.stdout.strip().splitlines()[-1]
    assert out.returncode == 0, out.stdout[-2000:]
    assert " mismatches 0 " in last + " ", last
    words = last.split()
    return int(words[1]), int(words[3])   # pairs checked, pairs skipped


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_near_misses_slivers_and_rays_in_the_plane(harness, seed):
    # single rays and narrow bundles that miss a triangle by 1e-7 .. 1e-1 of its size, rays in its plane, along its edges
    pairs, skipped = _run([harness, "random", "3000000", str(seed)])
    assert pairs > 2900000 and skipped > 50000   # (the test has to skip something to say anything)


@pytest.mark.parametrize("scene,w,h,stride,least", [("meshDragon", 3840, 2160, 13, 0.8), ("meshDragon", 640, 360, 3, 0.3),
                                                    ("meshMedBunny", 1920, 1080, 9, 0.5), ("meshTransparentBunny", 1280, 720, 7, 0.5)])
def test_tiles_of_real_frames(harness, scene, w, h, stride, least):
    # primary rays of 8x8 tiles and the shadow rays of their hit points, every triangle of every leaf they enter
    pairs, skipped = _run([harness, "scene", rt.asset("scenes/%s.json" % scene), os.path.join(rt.ASSETS, "resources.json"), rt.ASSETS,
                           str(w), str(h), str(stride)])
    assert skipped >= least * pairs, (pairs, skipped)


def test_the_harness_notices_a_skip_without_margin_and_guard(harness):
    # the same harness built with the geometric part of the test alone (no rounding margin, no guard against rays in a
    # triangle's plane) must find triangles that are skipped although the literal solve accepts them
    exe = EXE + "_nomargin"
    src = os.path.join(ROOT, "tests", "host_cull_check.cpp")
    libdir = os.path.join(ROOT, "raytracer_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-DRT_CULL_TEST_WITHOUT_MARGIN", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + libdir, "-lrt_amd", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe, "random", "1000000", "1"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
    os.remove(exe)
    last = out.stdout.strip().splitlines()[-1]
    assert out.returncode == 1 and int(last.split()[-1]) > 100, last
