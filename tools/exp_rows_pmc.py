#!/usr/bin/env python3
"""Renders bands of rows of the 4K meshDragon frame as launches of their own (to be run under rocprofv3 --pmc: the counters per
launch divided by the launch's waves say what a wave of that part of the picture costs - sky, floor, dragon).
   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d OUT -- python3 tools/exp_rows_pmc.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import raytracer_amd as rt

w, h = 3840, 2160
ds = rt.DeviceScene(rt.Scene("scenes/meshDragon.json", w, h, 3))
rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
for r0 in range(0, h, 120):
    st = ds.render_into(rgb.data_ptr(), r0, min(r0 + 120, h), want_stats=True)
    print("rows %d..%d: %d rays, %d shadow rays, kernel %.3f ms" % (r0, r0 + 120, st["rays"], st["shadow_rays"], st["kernel_ms"]))
