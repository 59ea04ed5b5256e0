#!/usr/bin/env python3
"""Kernel time, ray counts and wave-level iteration counts of mesh scenes at 512x512 for ray depths 0..3:
shows how much of a frame's time the bounce rays of a few tiles take."""
import sys; sys.path.insert(0,'.')
import raytracer_amd as rt
for scene in ["meshTransparentBunny","meshTorus","meshMedBunny"]:
    for depth in [0,1,2,3]:
        ds = rt.DeviceScene(rt.Scene("scenes/%s.json"%scene, 512, 512, depth))
        best=None
        for _ in range(3):
            _,_,_,st = ds.render_host(want_ids=False, flags=rt.RT_RENDER_TIME)
            best = st if best is None or st["kernel_ms"]<best["kernel_ms"] else best
        _,_,_,sc = ds.render_host(want_ids=False, flags=rt.RT_RENDER_COUNT_TESTS)
        print(scene, depth, "rays", best["rays"], "traced", best["rays_traced"], "ms %.3f"%best["kernel_ms"], "node tests/ray %.0f tri/ray %.0f"%(sc["node_tests"]/sc["rays"], sc["tri_tests"]/sc["rays"]), "wave traces", sc["wave_paths"][0], "node_it/trace %.0f tri_it/trace %.0f"%(sc["wave_paths"][1]/max(sc["wave_paths"][0],1), sc["wave_paths"][4]/max(sc["wave_paths"][0],1)), flush=True)
        ds.close()
