#!/usr/bin/env python3
"""What a wave costs besides its walks: the floor of the meshDragon scene without the dragon, at 3840x2160, with its lights taken
away one by one (to be run under rocprofv3 --pmc; the counters of a launch divided by its waves are what a wave of that
kind issues: a ray that misses everything, a hit shaded by the ambient light alone, one shadow ray more, two more).
   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d OUT -- python3 tools/exp_wave_overhead.py
   python3 tools/pmc_by_dispatch.py OUT"""
import json
import os
import sys
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import raytracer_amd as rt

w, h = 3840, 2160
point = {"__type__": "point_light", "position": [0, 10, 20], "color": [0.5, 0.5, 0.5]}
direc = {"__type__": "directional_light", "direction": [0, -1, 0], "color": [0.3, 0.3, 0.3]}
ambient = {"__type__": "ambient_light", "color": [0.2, 0.2, 0.2]}
floor = {"__type__": "mesh", "file_path": "meshes/large_plane.obj", "materials": ["white_lambert"]}
cam = {"__type__": "camera", "fov": 45.0, "position": [0, 20, 20], "up": [0, 1, 0], "target": [0, 0, 0]}
up = dict(cam, target=[0, 40, 0])
variants = [("looking away: every ray misses", up, [point, direc, ambient]),
            ("floor, ambient light only", cam, [ambient]),
            ("floor, ambient + point light", cam, [point, ambient]),
            ("floor, ambient + directional light", cam, [direc, ambient]),
            ("floor, all three lights", cam, [point, direc, ambient])]
rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
tmp = tempfile.mkdtemp()
for k, (name, camera, lights) in enumerate(variants):
    path = os.path.join(tmp, "variant%d.json" % k)
    json.dump({"__type__": "scene", "camera": camera, "objects": [floor], "lights": lights, "params": {"background_color": [0.4, 0.4, 0.4]}}, open(path, "w"))
    ds = rt.DeviceScene(rt.Scene(path, w, h, 3))
    for _ in range(2):
        st = ds.render_into(rgb.data_ptr(), want_stats=True)
    print("%d launches 2 | %s: %d rays, %d shadow rays, %d walks" % (k, name, st["rays"], st["shadow_rays"], st["rays_traced"]))
