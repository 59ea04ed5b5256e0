#!/usr/bin/env python3
"""How many of the triangles that the walks of a frame meet lie in leaves that only a few rays of the wave enter (diagnostic
build: RT_GEN_COUNT=leaf tools/build_variant.sh leafcount -DRT_WAVE_TIMES -DRT_WAVE_TIMES_CLK -DRT_STAGE_COUNT=2).
   RT_AMD_LIB=build/variants/librt_amd_leafcount.so python tools/exp_leaf_lanes.py [scene ...]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import tempfile
import raytracer_amd as rt
from raytracer_amd.synthetic import write_icosphere_scene

w, h = 3840, 2160
for name in (sys.argv[1:] or ["meshTransparentBunny", "meshDragon", "meshMedBunny", "icosphere:8"]):
    if name.startswith("icosphere:"):
        base = tempfile.mkdtemp()
        path, _ = write_icosphere_scene(base, int(name.split(":")[1]))
        sc = rt.Scene(path, w, h, 3, base_dir=base)
    else:
        sc = rt.Scene("scenes/%s.json" % name, w, h, 3)
    ds = rt.DeviceScene(sc)
    for flags, what in ((0, "default"), (rt.RT_RENDER_EVERY_TRIANGLE, "every triangle")):
        rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_TIME | flags)
        lf = ts.view(np.uint32).reshape(-1)[(1 << 22) + (1 << 21) + (1 << 20):][:5].astype(np.int64)
        n, n2, n4, nk, walks = (int(x) for x in lf)
        print("%-22s %-14s %d walks; triangles of the leaves entered %d; in leaves entered by at most 2 rays %.1f %%, at most 4 rays %.1f %%; rays per leaf entered (weighted by its triangles) %.1f" % (
            name, what, walks, n, 100.0 * n2 / max(n, 1), 100.0 * n4 / max(n, 1), nk / max(n, 1)))
