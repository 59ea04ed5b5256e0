#!/usr/bin/env python3
"""Bounce rays from queues (generation by generation) against bounce rays on each lane's stack: time and equality."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import raytracer_amd as rt

scenes = sys.argv[1:] or ["meshTransparentBunny", "meshTorus", "meshReflectedSquircle", "heliumAttenuatedSphere", "lambertBPHSpheres", "boxedSpheres"]
for size in ((512, 512), (1920, 1080), (3840, 2160)):
    for name in scenes:
        ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, size[0], size[1], 3))
        out = {}
        for label, fl in (("queued", rt.RT_RENDER_QUEUED), ("stacked", rt.RT_RENDER_STACKED)):
            best = 1e30
            for _ in range(4):
                rgb, ids, ts, st = ds.render_host(want_ids=False, flags=fl | rt.RT_RENDER_TIME)
                best = min(best, st["kernel_ms"])
            out[label] = (best, rgb, st["rays"])
        same = np.array_equal(out["queued"][1], out["stacked"][1])
        print("%-24s %4dx%-4d rays %9d  queued %8.3f ms  stacked %8.3f ms  identical %s" %
              (name, size[0], size[1], out["queued"][2], out["queued"][0], out["stacked"][0], same), flush=True)
        ds.close()
