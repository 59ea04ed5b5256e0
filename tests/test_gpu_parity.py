"""HIP render path (through the C ABI) against the CPU oracle on the same scenes.

Bar: primary closest-hit primitive ids and hit distances bit-exact; work counters of the instrumented
kernel identical to the oracle's (the traversal takes the same decisions node by node); final RGB8
within 1 LSB on at least 99.9 % of pixels (device powf/expf/pow differ from glibc's by ulps).
"""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import raytracer_amd as rt
from oracle_api import OracleScene
from scenes_common import ALL_SCENES, COUNTER_KEYS, REFERENCE

pytestmark = pytest.mark.gpu


def _rgb_close(a, b):
    diff = np.abs(a.astype(np.int16) - b.astype(np.int16)).max(axis=2)
    frac_ok = float((diff <= 1).mean())
    return frac_ok, int((diff > 0).sum()), int(diff.max())


def _check_scene(name, w, h, depth=3, base_dir=None):
    path = name if name.endswith(".json") else "scenes/%s.json" % name
    o = OracleScene(path, w, h, depth, base_dir=base_dir)
    orgb, oids, ots, oct_ = o.render(threads=8)
    ds = rt.DeviceScene(rt.Scene(path, w, h, depth, base_dir=base_dir))
    # instrumented variant: full traversals, counts every test
    rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_COUNT_TESTS)
    assert np.array_equal(ids, oids), "primary hit ids differ on %d pixels" % int((ids != oids).sum())
    assert np.array_equal(ts.view(np.uint32), ots.view(np.uint32)), "primary hit distances differ"
    frac_ok, n_diff, max_diff = _rgb_close(rgb, orgb)
    assert frac_ok >= 0.999, (name, frac_ok, n_diff, max_diff)
    for k in COUNTER_KEYS + ("ub_leave_level0", "ub_level_overflow", "ub_stack_overflow"):
        if name == "meshTransparentBunny" and k != "rays" and k != "shadow_rays":
            # Refraction directions differ by ulps between the device's and glibc's powf / pow / expf, so a few rays inside the
            # glass take another path: the work counts may drift, by a bounded amount that is printed (pytest -s) and held
            # here (measured drift: a few units in 10^5; the bound is 0.2 %, or 4 events for the two rare-event counters).
            drift = abs(st[k] - oct_[k])
            print("%s %dx%d %s: device %d oracle %d drift %d" % (name, w, h, k, st[k], oct_[k], drift))
            assert drift <= max(4, 0.002 * oct_[k]), (name, k, st[k], oct_[k])
            continue
        assert st[k] == oct_[k], (name, k, st[k], oct_[k])
    if name == "meshTransparentBunny":
        print("%s %dx%d: %d of %d pixels differ from the oracle (largest difference %d)" % (name, w, h, n_diff, w * h, max_diff))
        assert n_diff <= 0.001 * w * h
    # production variant (shadow rays stop at the first occluder, one shadow ray per hit point and light):
    # same image, same count of reference rays, fewer traversals
    rgb2, ids2, ts2, st2 = ds.render_host()
    assert np.array_equal(rgb2, rgb) and np.array_equal(ids2, ids) and np.array_equal(ts2.view(np.uint32), ts.view(np.uint32))
    assert st2["rays"] == st["rays"] and st2["shadow_rays"] == st["shadow_rays"]
    assert st2["rays_traced"] <= st["rays_traced"] == st["rays"]
    # bounce rays on each lane's own stack instead of generation by generation from queues: same bytes, same counts
    # (the library picks one of the two per scene; both are forced here)
    for way in (rt.RT_RENDER_STACKED, rt.RT_RENDER_QUEUED):
        rgb4, ids4, ts4, st4 = ds.render_host(flags=way)
        assert np.array_equal(rgb4, rgb2), "queued and stacked bounce rays differ on %d bytes" % int((rgb4 != rgb2).sum())
        assert np.array_equal(ids4, ids2)
        for k in ("rays", "shadow_rays", "rays_traced", "ub_leave_level0", "ub_level_overflow", "ub_stack_overflow"):
            assert st4[k] == st2[k], (name, k, st4[k], st2[k])
        rgb5, _, _, st5 = ds.render_host(want_ids=False, flags=way | rt.RT_RENDER_COUNT_TESTS)
        assert np.array_equal(rgb5, rgb)
        for k in COUNTER_KEYS:
            assert st5[k] == st[k], (name, k, st5[k], st[k])
    # cross-check kernel (one stack per pixel, the reference's solves evaluated as written): identical output
    rgb3, ids3, ts3, st3 = ds.render_host(flags=rt.RT_RENDER_LITERAL | rt.RT_RENDER_COUNT_TESTS)
    assert np.array_equal(rgb3, rgb) and np.array_equal(ids3, ids) and np.array_equal(ts3.view(np.uint32), ts.view(np.uint32))
    for k in COUNTER_KEYS:
        assert st3[k] == st[k], (name, k)
    return frac_ok, n_diff, max_diff


@pytest.mark.parametrize("name", ALL_SCENES)
def test_scene_parity_256(name):
    _check_scene(name, 256, 256)


def test_sphere_512_matches_reference_image_png():
    import os
    from PIL import Image
    from scenes_common import HERE
    ref = np.array(Image.open(os.path.join(HERE, "golden", "sphere_512_reference_image.png")).convert("RGB"))
    rgb, _, _, st = rt.DeviceScene(rt.Scene("scenes/sphere.json", 512, 512, 3)).render_host()
    assert np.array_equal(rgb, ref)
    assert st["rays"] == 512 * 512


@pytest.mark.parametrize("name,w,h", [("meshMedBunny", 333, 187), ("lambertBPHSpheres", 130, 75), ("sphere", 5, 3)])
def test_ragged_sizes(name, w, h):
    # sizes that are not multiples of the 16x16 workgroup tile, down to an image smaller than one tile
    _check_scene(name, w, h)


@pytest.mark.parametrize("depth", [0, 1, 6, 9])
def test_ray_depth_limits(depth):
    _check_scene("heliumAttenuatedSphere", 128, 128, depth)
    _check_scene("meshTorus", 96, 96, depth)


def test_row_tiles_reproduce_the_full_image():
    # multi-GPU partition: any split into row bands must give the same bytes as one full render
    sc = rt.Scene("scenes/meshMedBunny.json", 320, 200, 3)
    ds = rt.DeviceScene(sc)
    full, fids, _, fst = ds.render_host()
    for bands in ([0, 100, 200], [0, 7, 8, 33, 150, 200], [0, 25, 50, 75, 100, 125, 150, 175, 200]):
        rgb = np.zeros_like(full)
        ids = np.full_like(fids, -1)
        rays = 0
        for b, e in zip(bands[:-1], bands[1:]):
            r, i, _, st = ds.render_host(b, e)
            rgb[b:e] = r[b:e]
            ids[b:e] = i[b:e]
            assert not r[:b].any() and not r[e:].any()   # rows outside the band are untouched
            rays += st["rays"]
        assert np.array_equal(rgb, full) and np.array_equal(ids, fids)
        assert rays == fst["rays"]


def test_empty_band_and_bad_rows():
    ds = rt.DeviceScene(rt.Scene("scenes/sphere.json", 64, 64, 3))
    rgb, _, _, st = ds.render_host(10, 10)
    assert st["rays"] == 0 and not rgb.any()
    with pytest.raises(rt.RtError):
        ds.render_host(0, 65)
    with pytest.raises(rt.RtError):
        ds.render_host(-1, 4)


def test_device_pointer_entry_point_with_torch_stream():
    import torch
    sc = rt.Scene("scenes/lambertBPHSpheres.json", 256, 256, 3)
    ds = rt.DeviceScene(sc)
    ref, rids, _, _ = ds.render_host()
    rgb = torch.zeros((256, 256, 3), dtype=torch.uint8, device="cuda:0")
    ids = torch.zeros((256, 256), dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ds.render_into(rgb.data_ptr(), 0, 256, ids_ptr=ids.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    assert np.array_equal(rgb.cpu().numpy(), ref) and np.array_equal(ids.cpu().numpy(), rids)


def test_dragon_4k_ray_count_and_band_parity():
    # Headline configuration.  Whole-frame ray count must equal the reference run's; a 96-row band
    # through the dragon is compared with the oracle pixel by pixel.
    exp = REFERENCE["meshDragon_3840x2160"]
    sc = rt.Scene("scenes/meshDragon.json", 3840, 2160, 3)
    ds = rt.DeviceScene(sc)
    rgb, ids, ts, st = ds.render_host()
    assert st["rays"] == exp["rays"] and st["shadow_rays"] == exp["shadow_rays"]
    # The whole 8.3 M-pixel frame is byte-identical to the frame of the reference's own routines (digest recorded from
    # the reference run; this scene has no dielectric, so no device/glibc libm difference reaches a pixel) ...
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == exp["rgb_sha256"]
    # ... and the instrumented variant makes exactly the reference's number of node, leaf, triangle and sphere tests.
    rgb_c, _, _, st_c = ds.render_host(want_ids=False, flags=rt.RT_RENDER_COUNT_TESTS)
    assert np.array_equal(rgb_c, rgb)
    for k in COUNTER_KEYS:
        assert st_c[k] == exp[k], (k, st_c[k], exp[k])
    o = OracleScene("scenes/meshDragon.json", 3840, 2160, 3)
    b, e = 1000, 1096
    orgb, oids, ots, _ = o.render(b, e, threads=16)
    assert np.array_equal(ids[b:e], oids[b:e])
    assert np.array_equal(ts[b:e].view(np.uint32), ots[b:e].view(np.uint32))
    frac_ok, n_diff, max_diff = _rgb_close(rgb[b:e], orgb[b:e])
    assert frac_ok >= 0.999, (frac_ok, n_diff, max_diff)


@pytest.mark.parametrize("scene", ["meshMedBunny", "meshTransparentBunny"])   # (the second: bounce rays from queues, pixels named by their place in the band buffer)
@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_band_partition_reproduces_the_full_image(nranks, scene):
    import torch
    w, h = 200, 150   # 10 bands of 16 rows, the last one ragged
    ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % scene, w, h, 3))
    full, fids, _, fst = ds.render_host()
    tiles, id_tiles, rays = [], [], 0
    nb_max = ds.band_count(16, 0, nranks)
    for r in range(nranks):
        buf = torch.zeros((nb_max * 16, w, 3), dtype=torch.uint8, device="cuda:0")
        ibuf = torch.full((nb_max * 16, w), -7, dtype=torch.int32, device="cuda:0")
        st = ds.render_bands_into(buf.data_ptr(), 16, r, nranks, ids_ptr=ibuf.data_ptr(), want_stats=True)
        rays += st["rays"]
        tiles.append(buf.cpu().numpy())
        id_tiles.append(ibuf.cpu().numpy())
    assert np.array_equal(rt.assemble_bands(tiles, h, 16), full)
    assert np.array_equal(rt.assemble_bands(id_tiles, h, 16), fids)
    assert rays == fst["rays"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["meshMedBunny", "meshTransparentBunny"])
def test_two_frames_in_flight_on_two_streams(name):
    # the pipelining bench.py uses with several GPUs: one uploaded scene per stream, launches overlapping on the device;
    # every frame must come out with the bytes of a frame rendered alone
    import torch
    w, h = 320, 208
    scene = rt.Scene("scenes/%s.json" % name, w, h, 3)
    scenes = [rt.DeviceScene(scene), rt.DeviceScene(scene)]
    alone, _, _, _ = scenes[0].render_host(want_ids=False)
    streams = [torch.cuda.Stream(device="cuda:0"), torch.cuda.Stream(device="cuda:0")]
    bufs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()
    for k, buf in enumerate(bufs):
        with torch.cuda.stream(streams[k % 2]):
            scenes[k % 2].render_bands_into(buf.data_ptr(), 16, 0, 1, stream=streams[k % 2].cuda_stream,
                                            flags=rt.RT_RENDER_STRIPS if k >= 2 else 0)   # (tiles dealt to the dies in strips, as the bench asks)
    torch.cuda.synchronize()
    for buf in bufs:
        assert np.array_equal(buf.cpu().numpy(), alone)


@pytest.mark.parametrize("name,w,h", [("meshMedBunny", 333, 187), ("lambertBPHSpheres", 130, 75), ("sphere", 5, 3),
                                      ("meshTorus", 1000, 40), ("meshMedBunny", 8, 600), ("twoSpheres", 264, 16)])
def test_tile_strips_on_ragged_sizes(name, w, h):
    # tiles dealt to the dies in strips of four tile columns: widths that leave the last strip, or whole dies, without
    # tiles (1, 2, 17, 33, 42 and 125 tile columns) must give the bytes of the plain grid, in full frames and in bands
    ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
    rgb, ids, ts, st = ds.render_host()
    rgb2, ids2, ts2, st2 = ds.render_host(flags=rt.RT_RENDER_STRIPS)
    assert np.array_equal(rgb2, rgb) and np.array_equal(ids2, ids) and np.array_equal(ts2.view(np.uint32), ts.view(np.uint32))
    assert st2["rays"] == st["rays"] and st2["rays_traced"] == st["rays_traced"]
    b0, b1 = h // 3, max(h // 3 + 1, (2 * h) // 3)
    part, pids, _, _ = ds.render_host(b0, b1, flags=rt.RT_RENDER_STRIPS)
    assert np.array_equal(part[b0:b1], rgb[b0:b1]) and np.array_equal(pids[b0:b1], ids[b0:b1])
    assert not part[:b0].any() and not part[b1:].any()


@pytest.mark.parametrize("name,w,h", [("sphere", 65535, 3), ("lambertBPHSpheres", 3, 65535), ("meshMedBunny", 65535, 2)])
def test_longest_image_sides(name, w, h):
    # the reference's image sides are unsigned shorts: the longest row and the longest column against the oracle, on
    # the plain grid and with the tiles dealt to the dies in strips (8 192 tile columns, 2 048 strips)
    o = OracleScene("scenes/%s.json" % name, w, h, 3)
    orgb, oids, ots, _ = o.render(threads=8)
    ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
    for flags in (0, rt.RT_RENDER_STRIPS):
        rgb, ids, ts, st = ds.render_host(flags=flags)
        assert np.array_equal(ids, oids) and np.array_equal(ts.view(np.uint32), ots.view(np.uint32))
        assert _rgb_close(rgb, orgb)[0] >= 0.999
        assert st["rays"] >= w * h


_DENSE_CHILD = """
import hashlib, sys
import numpy as np
import raytracer_amd as rt
for name, w, h in [("meshDragon", 333, 187), ("meshMedBunny", 520, 96), ("meshDragon", 64, 64)]:
    ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
    for flags in (0, rt.RT_RENDER_STRIPS):
        rgb, ids, ts, st = ds.render_host(flags=flags)
        print(name, w, h, hashlib.sha256(rgb.tobytes() + ids.tobytes() + ts.tobytes()).hexdigest(), st["rays"], st["rays_traced"])
"""


def test_six_waves_per_simd_variant_on_small_and_ragged_frames():
    # The variant of the plain kernel compiled for six waves per SIMD is chosen for big frames of big meshes only (the
    # 4K dragon test runs it); the threshold is read once per process, so a child process with the threshold at one
    # tile and 8x8 tiles forced renders small ragged frames with it: same bytes, ids, distances and ray counts.
    import os
    import subprocess
    import sys
    from scenes_common import HERE
    want = []
    for name, w, h in [("meshDragon", 333, 187), ("meshMedBunny", 520, 96), ("meshDragon", 64, 64)]:
        ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
        rgb, ids, ts, st = ds.render_host()
        for _ in range(2):
            want.append("%s %d %d %s %d %d" % (name, w, h, hashlib.sha256(rgb.tobytes() + ids.tobytes() + ts.tobytes()).hexdigest(),
                                               st["rays"], st["rays_traced"]))
    env = dict(os.environ, RT_AMD_DENSE_MIN_TILES="1", RT_AMD_TILE="8")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(HERE, ".."), HERE, env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", _DENSE_CHILD], env=env, capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines() == want


SPOT_SCENE = """{
  "camera": {"fov": 50.0, "position": [0, 6, 14], "up": [0, 1, 0], "target": [0, 1, 0]},
  "objects": [
    {"__type__": "mesh", "file_path": "meshes/large_plane.obj", "materials": ["white_lambert", "white_bph_100"]},
    {"__type__": "mesh", "file_path": "meshes/sphere.obj", "materials": ["copper_lambert", "copper_bph_100"],
     "scaling": [2, 2, 2], "rotation": [10, 20, 30], "translation": [-2, 2, 0]},
    {"__type__": "sphere", "position": [3, 1.5, 1], "radius": 1.5, "materials": ["blue_lambert", "blue_bph_1000"]}
  ],
  "lights": [
    {"__type__": "spot_light", "position": [0, 9, 4], "color": [0.9, 0.9, 0.8], "angle": 55.0, "direction": [0, -1, -0.3]},
    {"__type__": "spot_light", "position": [-6, 5, 6], "color": [0.3, 0.5, 0.9], "angle": 30.0, "direction": [0.7, -0.5, -0.7]},
    {"__type__": "point_light", "position": [5, 8, 5], "color": [0.2, 0.2, 0.2]},
    {"__type__": "ambient_light", "color": [0.1, 0.1, 0.1]}
  ],
  "params": {"background_color": [0.1, 0.2, 0.3], "enable_shadows": "true"}
}"""


def test_spot_lights_synthetic_scene(tmp_path):
    # No scene of the reference uses a spot light, although its routines implement one (lights.cu:281-295,335-350).
    p = tmp_path / "spots.json"
    p.write_text(SPOT_SCENE)
    for shadows in ("true", "false"):
        q = tmp_path / ("spots_%s.json" % shadows)
        q.write_text(SPOT_SCENE.replace('"enable_shadows": "true"', '"enable_shadows": "%s"' % shadows))
        _check_scene(str(q), 200, 120)


PARALLEL_LIGHTS_SCENE = """{
  "camera": {"fov": 45.0, "position": [3, 9, 12], "up": [0, 1, 0], "target": [0, 1.5, 0]},
  "objects": [
    {"__type__": "mesh", "file_path": "meshes/large_plane.obj", "materials": ["white_lambert"]},
    {"__type__": "mesh", "file_path": "meshes/bunny_med.obj", "materials": ["red_lambert", "white_bph_1000"],
     "scaling": [2, 2, 2], "rotation": [0, 30, 0], "translation": [0, 0.2, 0]}
  ],
  "lights": [
    {"__type__": "directional_light", "direction": [0, -1, 0], "color": [0.2, 0.2, 0.2]},
    {"__type__": "directional_light", "direction": [-1, 0, 0], "color": [0.2, 0.1, 0.1]},
    {"__type__": "directional_light", "direction": [0, 0, -1], "color": [0.1, 0.2, 0.1]},
    {"__type__": "directional_light", "direction": [-1, -1, 0], "color": [0.1, 0.1, 0.2]},
    {"__type__": "directional_light", "direction": [0, -1, -1], "color": [0.2, 0.2, 0.1]},
    {"__type__": "directional_light", "direction": [1, -1, 0], "color": [0.1, 0.1, 0.1]},
    {"__type__": "directional_light", "direction": [-1, 0, -1], "color": [0.1, 0.2, 0.2]},
    {"__type__": "ambient_light", "color": [0.1, 0.1, 0.1]}
  ],
  "params": {"background_color": [0.3, 0.3, 0.4], "enable_shadows": "true"}
}"""


def test_lights_parallel_to_axes_and_planes(tmp_path):
    # Shadow rays with one or two direction components exactly zero, in every combination: each zero pattern has a scan
    # loop of its own in the wave kernel.
    p = tmp_path / "parallel_lights.json"
    p.write_text(PARALLEL_LIGHTS_SCENE)
    _check_scene(str(p), 224, 136)


def test_lights_with_direction_components_too_small_to_invert(tmp_path):
    # Shadow rays with a direction component that is not zero but so small that 1 / d overflows (and one whose inverse is
    # merely huge): the plane distances of the conservative node classification cannot be formed for them, every node test
    # of such a ray has to come from the exact, division-free form.  A mirror variant sends the same rays through the walk
    # of the kernels with bounce rays.
    lights = '''
    {"__type__": "directional_light", "direction": [1e-40, -1, 0], "color": [0.3, 0.2, 0.2]},
    {"__type__": "directional_light", "direction": [0, -1, -3e-39], "color": [0.2, 0.3, 0.2]},
    {"__type__": "directional_light", "direction": [2e-39, -1, 4e-41], "color": [0.2, 0.2, 0.3]},
    {"__type__": "directional_light", "direction": [1e-30, -1, -1e-33], "color": [0.1, 0.1, 0.1]},
    {"__type__": "ambient_light", "color": [0.1, 0.1, 0.1]}'''
    import re
    scene = re.sub(r'"lights": \[.*?\n  \],', lambda m: '"lights": [' + lights + "\n  ],", PARALLEL_LIGHTS_SCENE, flags=re.S)
    p = tmp_path / "tiny_components.json"
    p.write_text(scene)
    _check_scene(str(p), 224, 136)
    q = tmp_path / "tiny_components_mirror.json"
    q.write_text(scene.replace('["white_lambert"]', '["white_lambert", "gray_mirror"]'))
    _check_scene(str(q), 160, 96)


@pytest.mark.parametrize("k,w,h", [(3, 256, 256), (5, 320, 180)])
def test_synthetic_icosphere_scene(tmp_path, k, w, h):
    # The benchmark's synthetic mesh (20 * 4**k triangles over the large plane), at sizes the oracle renders in seconds.
    from raytracer_amd.synthetic import write_icosphere_scene
    path, n = write_icosphere_scene(str(tmp_path), k)
    assert n == 20 * 4 ** k
    _check_scene(path, w, h, base_dir=str(tmp_path))


def test_forked_contexts_share_one_upload_and_render_concurrently():
    # rt_dscene_fork: a second render context (counters, queues) on the scene arrays already in HBM; frames in flight on
    # two streams come out with the bytes of a frame rendered alone, and the arrays outlive the handle that uploaded them
    import torch
    w, h = 320, 208
    for name in ("meshMedBunny", "meshTransparentBunny"):
        first = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
        alone, _, _, st_alone = first.render_host(want_ids=False)
        second = first.fork()
        streams = [torch.cuda.Stream(device="cuda:0"), torch.cuda.Stream(device="cuda:0")]
        bufs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
        torch.cuda.synchronize()
        for k, buf in enumerate(bufs):
            with torch.cuda.stream(streams[k % 2]):
                (first, second)[k % 2].render_into(buf.data_ptr(), stream=streams[k % 2].cuda_stream)
        torch.cuda.synchronize()
        for buf in bufs:
            assert np.array_equal(buf.cpu().numpy(), alone)
        first.close()                       # the fork keeps the arrays alive
        rgb, _, _, st = second.render_host(want_ids=False)
        assert np.array_equal(rgb, alone) and st["rays"] == st_alone["rays"]


def test_device_prepare_is_optional_and_repeatable():
    L = rt.lib()
    assert L.rt_device_prepare(0) == 0 and L.rt_device_prepare(0) == 0
    assert L.rt_device_prepare(-1) == -1 and L.rt_device_prepare(1 << 20) == -1
    sc = rt.Scene("scenes/sphere.json", 64, 64, 3)
    rgb = rt.DeviceScene(sc).render_host()[0]
    assert rgb.shape == (64, 64, 3) and rgb.any()


def test_entry_points_leave_the_current_device_alone():
    import torch
    before = torch.cuda.current_device()
    ds = rt.DeviceScene(rt.Scene("scenes/sphere.json", 64, 64, 3), device=0)
    ds.render_host()
    ds.close()
    assert torch.cuda.current_device() == before


@pytest.mark.parametrize("name,w,h", [("meshMedBunny", 200, 150), ("meshTransparentBunny", 160, 96), ("sphere", 5, 3)])
def test_gathered_frame_on_the_visible_gpus(name, w, h):
    # The multi-GPU product path (run --gpus N): bands on every visible device of this process, one RCCL gather, unpack on
    # the first device.  On a one-GPU box N = 1 (the gather is rank 0 sending to itself); the frame must be rt_render_rows_host's.
    import torch
    n = torch.cuda.device_count()
    scene = rt.Scene("scenes/%s.json" % name, w, h, 3)
    dss = [rt.DeviceScene(scene, device=r) for r in range(n)]
    comms = rt.Comm.init_all(list(range(n)))
    want, _, _, st_want = dss[0].render_host(want_ids=False)
    for _ in range(2):
        rgb, st = rt.render_frame_gathered(dss, comms)
        assert np.array_equal(rgb, want)
        assert st["rays"] == st_want["rays"] and st["shadow_rays"] == st_want["shadow_rays"]
    for c in comms:
        c.close()


def test_gather_and_unpack_entry_points_with_a_world_of_one():
    # rt_comm_unique_id / rt_comm_init_rank / rt_gather_bands / rt_unpack_bands as one process per GPU would call them,
    # here with one rank; the unpacking of 1, 2, 3 and 8 ranks' tiles is checked against the host-side assembly.
    import torch
    w, h = 200, 150
    ds = rt.DeviceScene(rt.Scene("scenes/meshMedBunny.json", w, h, 3))
    full, fids, _, _ = ds.render_host()
    comm = rt.Comm.init_rank(rt.comm_unique_id(), 1, 0, 0)
    for nranks in (1, 2, 3, 8):
        nb_max = ds.band_count(16, 0, nranks)
        tile_bytes = nb_max * 16 * w * 3
        gathered = torch.zeros((nranks, nb_max * 16, w, 3), dtype=torch.uint8, device="cuda:0")
        gids = torch.zeros((nranks, nb_max * 16, w), dtype=torch.int32, device="cuda:0")
        for r in range(nranks):
            local = torch.zeros((nb_max * 16, w, 3), dtype=torch.uint8, device="cuda:0")
            ds.render_bands_into(local.data_ptr(), 16, r, nranks, ids_ptr=gids[r].data_ptr())
            # a world of one: the "gather" moves the tile into rank r's slot of the gathered buffer
            comm.gather_bands(local.data_ptr(), tile_bytes, gathered[r].data_ptr())
            torch.cuda.synchronize()
        image = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
        rt.unpack_bands(gathered.data_ptr(), tile_bytes, nranks, 16, w, h, 3, image.data_ptr())
        idimg = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
        rt.unpack_bands(gids.data_ptr(), nb_max * 16 * w * 4, nranks, 16, w, h, 4, idimg.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(image.cpu().numpy(), full)
        assert np.array_equal(idimg.cpu().numpy(), fids)
        assert np.array_equal(rt.assemble_bands([g.cpu().numpy() for g in gathered], h, 16), full)
    comm.close()


@pytest.mark.parametrize("name", ALL_SCENES)
def test_against_committed_golden_fixtures(name):
    # tests/golden/scenes_128: ids, distances, RGB and counters of the CPU oracle as rendered where the fixtures were made
    # (tools/make_goldens.py).  They are the oracle's own output - a guard against drift between machines and builds, not a
    # second pin of the oracle: what pins the oracle is tests/test_oracle_pins.py.  The HIP path is held to them directly, and so is the oracle as built on this machine:
    # a drift of the oracle or of the host's libm between machines shows here instead of cancelling out.
    import json
    import os
    from scenes_common import HERE
    gdir = os.path.join(HERE, "golden", "scenes_128")
    with open(os.path.join(gdir, "manifest.json")) as f:
        man = json.load(f)["scenes"][name]
    g = np.load(os.path.join(gdir, name + ".npz"))
    w, h, depth = man["width"], man["height"], man["depth"]
    ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, depth))
    rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_COUNT_TESTS)
    assert np.array_equal(ids, g["ids"]), "primary hit ids differ from the fixture on %d pixels" % int((ids != g["ids"]).sum())
    assert np.array_equal(ts.view(np.uint32), g["t_bits"]), "primary hit distances differ from the fixture"
    frac_ok, n_diff, max_diff = _rgb_close(rgb, g["rgb"])
    assert frac_ok >= 0.999, (name, frac_ok, n_diff, max_diff)
    # the production variant (assembly walk, shadow rays stopped at their first blocker, leaf triangles ruled out per wave) is
    # held to the same fixture directly, not only through its equality with the instrumented variant elsewhere in this file
    rgb_p, ids_p, ts_p, _ = ds.render_host()
    assert np.array_equal(ids_p, g["ids"]) and np.array_equal(ts_p.view(np.uint32), g["t_bits"]), "production variant: ids / distances differ from the fixture"
    p_ok, p_diff, p_max = _rgb_close(rgb_p, g["rgb"])
    assert p_ok >= 0.999, ("production variant", name, p_ok, p_diff, p_max)
    if name != "meshTransparentBunny":
        for k in COUNTER_KEYS:
            assert st[k] == man["counters"][k], (name, k, st[k], man["counters"][k])
    orgb, oids, ots, oct_ = OracleScene("scenes/%s.json" % name, w, h, depth).render(threads=8)
    assert np.array_equal(oids, g["ids"]) and np.array_equal(ots.view(np.uint32), g["t_bits"]), "the oracle built here left its fixture"
    o_ok, o_diff, o_max = _rgb_close(orgb, g["rgb"])
    assert o_ok >= 0.999, ("oracle vs fixture", name, o_diff, o_max)


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
def test_glass_bunny_full_size_both_ways(w, h):
    # the configuration rendered from queues, at the sizes it is measured at: bounce rays generation by generation from
    # queues and on the lanes' stacks give the same bytes and the same ray counts (a property; the oracle takes minutes here)
    ds = rt.DeviceScene(rt.Scene("scenes/meshTransparentBunny.json", w, h, 3))
    a, ida, _, sa = ds.render_host(flags=rt.RT_RENDER_QUEUED)
    b, idb, _, sb = ds.render_host(flags=rt.RT_RENDER_STACKED)
    c, _, _, sc = ds.render_host(want_ids=False)           # the library's own choice (queues)
    assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(ida, idb)
    for k in ("rays", "shadow_rays", "rays_traced", "ub_leave_level0", "ub_level_overflow", "ub_stack_overflow"):
        assert sa[k] == sb[k] == sc[k], (k, sa[k], sb[k], sc[k])
    assert sa["rays"] > w * h and sa["from_queues"] == 1 and sc["from_queues"] == 1 and sb["from_queues"] == 0


def test_queue_overflow_falls_back_to_the_stacks(monkeypatch):
    # queues and block pool sized as for a 64-pixel call: they overflow at once, the frame is rendered on the stacks
    # instead, and nothing of the abandoned attempt shows (bytes and counts of the stacked way)
    ds = rt.DeviceScene(rt.Scene("scenes/meshTransparentBunny.json", 256, 256, 3))
    want, _, _, sw = ds.render_host(want_ids=False, flags=rt.RT_RENDER_STACKED)
    monkeypatch.setenv("RT_AMD_QUEUE_CAP_PIXELS", "64")
    got, _, _, sg = ds.render_host(want_ids=False, flags=rt.RT_RENDER_QUEUED)
    monkeypatch.delenv("RT_AMD_QUEUE_CAP_PIXELS")
    again, _, _, sa = ds.render_host(want_ids=False, flags=rt.RT_RENDER_QUEUED)      # and with room again: from queues
    assert np.array_equal(got, want) and np.array_equal(again, want)
    assert (sw["from_queues"], sw["queue_fallback"]) == (0, 0) and (sg["from_queues"], sg["queue_fallback"]) == (0, 1)
    assert (sa["from_queues"], sa["queue_fallback"]) == (1, 0)
    for k in ("rays", "shadow_rays", "rays_traced"):
        assert sg[k] == sw[k] == sa[k], (k, sg[k], sw[k], sa[k])



@pytest.mark.gpu
@pytest.mark.parametrize("name", ["meshDragon", "meshMedBunny", "meshTorus", "meshReflectedSquircle", "meshTransparentBunny"])
@pytest.mark.parametrize("w,h", [(333, 187), (1280, 720)])
def test_skipping_triangles_changes_nothing(name, w, h):
    # the wave-level triangle skip against RT_RENDER_EVERY_TRIANGLE (octtree.cu:264-274 as written): same bytes, same
    # hit ids and distances, same ray counts - on 4x4 and 8x8 tiles, ragged edges, bounce rays on the stacks and from queues
    ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
    ways = [0, rt.RT_RENDER_STACKED] if name == "meshTransparentBunny" else [0]
    for way in ways:
        a, ida, ta, sa = ds.render_host(flags=way)
        b, idb, tb, sb = ds.render_host(flags=way | rt.RT_RENDER_EVERY_TRIANGLE)
        assert np.array_equal(a, b) and np.array_equal(ida, idb) and np.array_equal(ta.view(np.uint32), tb.view(np.uint32))
        for k in ("rays", "shadow_rays", "rays_traced"):
            assert sa[k] == sb[k], (k, sa[k], sb[k])


_CULL_CHILD = """
import hashlib, sys
import numpy as np
import raytracer_amd as rt
from scenes_common import ALL_SCENES
for name in ALL_SCENES:
    for w, h in ((128, 128), (333, 187)):
        ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
        rgb, ids, ts, st = ds.render_host()
        print(name, w, h, hashlib.sha256(rgb.tobytes() + ids.tobytes() + ts.tobytes()).hexdigest(), st["rays"], st["rays_traced"])
"""


@pytest.mark.gpu
def test_skipping_triangles_on_small_meshes_too():
    # Meshes below 512 nodes are rendered without the triangle skip (it does not pay there); the limit is read once per
    # process, so a child process with the limit at zero renders every scene with it - boxes and planes made of a few huge
    # triangles, walls that shadow rays run along - and must produce the bytes, ids, distances and ray counts of this one.
    import os
    import subprocess
    import sys
    from scenes_common import HERE
    want = []
    for name in ALL_SCENES:
        for w, h in ((128, 128), (333, 187)):
            ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
            rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_EVERY_TRIANGLE)
            want.append("%s %d %d %s %d %d" % (name, w, h, hashlib.sha256(rgb.tobytes() + ids.tobytes() + ts.tobytes()).hexdigest(),
                                               st["rays"], st["rays_traced"]))
    env = dict(os.environ, RT_AMD_CULL_MIN_NODES="0")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(HERE, ".."), HERE, env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", _CULL_CHILD], env=env, capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines() == want


@pytest.mark.gpu
def test_skipping_triangles_from_random_viewpoints(tmp_path):
    # meshDragon and the glass bunny seen from seeded random viewpoints (far, close, grazing along the floor, from below the
    # horizon line of the mesh) under random point and directional lights: with and without the triangle skip the same
    # bytes, ids, distances and ray counts
    import json
    import os
    rng = np.random.RandomState(20261112)
    assets = rt.ASSETS
    os.symlink(os.path.join(assets, "meshes"), str(tmp_path / "meshes"))
    os.symlink(os.path.join(assets, "resources.json"), str(tmp_path / "resources.json"))
    os.mkdir(str(tmp_path / "scenes"))
    seen = 0
    for name, n_views in (("meshDragon", 10), ("meshTransparentBunny", 4)):
        with open(os.path.join(assets, "scenes", name + ".json")) as f:
            scene = json.load(f)
        for view in range(n_views):
            sc = json.loads(json.dumps(scene))
            dist = float(rng.choice([4.0, 9.0, 18.0, 35.0]))
            az, el = rng.uniform(0, 2 * np.pi), rng.choice([0.02, 0.15, 0.5, 1.1])
            target = [float(rng.uniform(-2, 2)), float(rng.uniform(0.5, 6)), float(rng.uniform(-2, 2))]
            sc["camera"]["position"] = [target[0] + dist * np.cos(el) * np.cos(az), target[1] + dist * np.sin(el), target[2] + dist * np.cos(el) * np.sin(az)]
            sc["camera"]["target"] = target
            sc["camera"]["fov"] = float(rng.choice([20.0, 45.0, 80.0]))
            for l in sc["lights"]:
                if l["__type__"] == "point_light":
                    l["position"] = [float(rng.uniform(-15, 15)), float(rng.uniform(1, 20)), float(rng.uniform(-15, 15))]
                if l["__type__"] == "directional_light":
                    d = rng.normal(size=3); d[1] = -abs(d[1]) - 0.05
                    l["direction"] = [float(x) for x in d]
            path = str(tmp_path / "scenes" / ("%s_view%d.json" % (name, view)))
            with open(path, "w") as f:
                json.dump(sc, f)
            w, h = [(640, 360), (333, 187), (1280, 720)][view % 3]
            ds = rt.DeviceScene(rt.Scene(path, w, h, 3, base_dir=str(tmp_path)))
            a, ida, ta, sa = ds.render_host()
            b, idb, tb, sb = ds.render_host(flags=rt.RT_RENDER_EVERY_TRIANGLE)
            assert np.array_equal(a, b) and np.array_equal(ida, idb) and np.array_equal(ta.view(np.uint32), tb.view(np.uint32)), (name, view)
            for k in ("rays", "shadow_rays", "rays_traced"):
                assert sa[k] == sb[k], (name, view, k, sa[k], sb[k])
            seen += int((ida >= 0).mean() > 0.05)   # (a twentieth of the pixels or more show a triangle)
            ds.close()
    assert seen >= 10


def test_queue_launches_sized_by_the_frame_before():
    # The launches that render a queue have waves for the longest generation of the context's last frame and a margin.  A frame
    # whose generations outgrow that is noticed on the device and rendered again with launches for full queues: same bytes, ids
    # and counts as on a fresh context, whatever the context rendered before.
    w, h = 1920, 1080
    sc = rt.Scene("scenes/meshTransparentBunny.json", w, h, 3)
    fresh = rt.DeviceScene(sc)
    want, wid, _, sw = fresh.render_host(flags=rt.RT_RENDER_QUEUED)
    assert sw["from_queues"] == 1 and sw["rays"] > w * h + 200000
    ds = rt.DeviceScene(sc)
    rows = np.flatnonzero((wid >= 0).any(axis=1))
    few, fid, _, sf = ds.render_host(int(rows[0]), int(rows[0]) + 2, flags=rt.RT_RENDER_QUEUED)      # the bunny's topmost rows: a handful of bounce rays
    assert sf["from_queues"] == 1 and np.array_equal(few[rows[0]:rows[0] + 2], want[rows[0]:rows[0] + 2]) and sf["rays"] > 2 * w
    got, gid, _, sg = ds.render_host(flags=rt.RT_RENDER_QUEUED)        # launches sized for that handful: too short
    again, aid, _, sa = ds.render_host(flags=rt.RT_RENDER_QUEUED)      # sized by the whole frame: enough
    assert (sg["from_queues"], sa["from_queues"]) == (2, 1)
    assert np.array_equal(got, want) and np.array_equal(again, want) and np.array_equal(gid, wid) and np.array_equal(aid, wid)
    for k in ("rays", "shadow_rays", "rays_traced"):
        assert sg[k] == sw[k] == sa[k], (k, sg[k], sw[k], sa[k])


def test_the_order_of_a_queue_never_shows_in_the_image():
    # The order in which a generation's rays are rendered is a matter of speed only: every ray writes its own term slot.  The
    # settings are read once per process, so each one renders in a process of its own; the digest of bytes, ids and counts
    # must be the one of the default order (direction class and 16^3 cells), also without any order, with the octant order
    # (whose keys use all 2 M bins where the default uses 393 216) and with other grids.
    code = ("import hashlib, raytracer_amd as rt\n"
            "ds = rt.DeviceScene(rt.Scene('scenes/meshTransparentBunny.json', 640, 360, 3))\n"
            "for _ in range(2):\n"
            "    rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_QUEUED)\n"
            "    assert st['from_queues'] == 1\n"
            "    print(hashlib.sha256(rgb.tobytes() + ids.tobytes() + ts.tobytes()).hexdigest(), st['rays'], st['shadow_rays'])\n")
    seen = {}
    for setting in ({}, {"RT_AMD_SORT_RAYS": "0"}, {"RT_AMD_SORT_DIRGRID": "0"}, {"RT_AMD_SORT_DIRGRID": "6", "RT_AMD_SORT_CELLBITS": "3"},
                    {"RT_AMD_SORT_DIRGRID": "2", "RT_AMD_SORT_CELLBITS": "5"}, {"RT_AMD_QUEUE_FULL_LAUNCHES": "1"}, {"RT_AMD_QUEUE_SMALL": "1000"}):
        env = dict(os.environ, **setting)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert out.returncode == 0, (setting, out.stderr[-2000:])
        lines = [l for l in out.stdout.splitlines() if len(l.split()) == 3]
        assert len(lines) == 2 and lines[0] == lines[1], (setting, out.stdout)
        seen[tuple(sorted(setting.items()))] = lines[0]
    assert len(set(seen.values())) == 1, seen
