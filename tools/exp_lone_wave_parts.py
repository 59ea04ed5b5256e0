#!/usr/bin/env python3
"""What the life of the longest wave of the headline frame is made of: tile row 103 of the 4K meshDragon frame rendered alone
(8x8 tiles forced, RT_AMD_TILE=8; diagnostic build -DRT_WAVE_TIMES selected with RT_AMD_LIB) with parts of the scene's lights
taken away, with and without the triangle skip.
   RT_AMD_TILE=8 RT_AMD_LIB=build/variants/librt_amd_times.so python tools/exp_lone_wave_parts.py"""
import json
import os
import sys
import tempfile
sys.path.insert(0, '.')
import numpy as np
import torch
import raytracer_amd as rt

w, h, ty = 3840, 2160, 103
tx = w // 8
base = json.load(open(rt.asset("scenes/meshDragon.json")))
rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
ts = torch.zeros((h, w), dtype=torch.float32, device="cuda")
tmp = tempfile.mkdtemp()


def variant(name, lights, shadows=True):
    sc = json.loads(json.dumps(base))
    sc["lights"] = [l for l in sc["lights"] if l["__type__"] in lights]
    sc["params"]["enable_shadows"] = "true" if shadows else "false"
    p = os.path.join(tmp, name + ".json")
    json.dump(sc, open(p, "w"))
    return p


cases = [("all lights", variant("all", ("point_light", "directional_light", "ambient_light"))),
         ("no shadow rays", variant("nosh", ("point_light", "directional_light", "ambient_light"), shadows=False)),
         ("ambient only", variant("amb", ("ambient_light",))),
         ("point light only", variant("pt", ("point_light",))),
         ("directional light only", variant("dir", ("directional_light",)))]
for what, path in cases:
    ds = rt.DeviceScene(rt.Scene(path, w, h, 3, base_dir=rt.ASSETS))
    for flags, fl in ((0, "skip"), (rt.RT_RENDER_EVERY_TRIANGLE, "every triangle")):
        for _ in range(2):
            ts.zero_()
            st = ds.render_into(rgb.data_ptr(), ty * 8, ty * 8 + 8, t_ptr=ts.data_ptr(), flags=rt.RT_RENDER_TIME | flags, want_stats=True)
        raw = ts.cpu().numpy().view(np.uint32).reshape(-1)[:4 * tx].reshape(tx, 4)
        start = raw[:, 0].astype(np.int64); end = raw[:, 1].astype(np.int64)
        end = np.where(end < start, end + (1 << 32), end)
        dur = (end - start) / 100.0
        clk = ""
        if os.environ.get("RT_EXP_CLK"):   # (build with -DRT_WAVE_TIMES_CLK: word 2 of a wave's record is its life in shader clock cycles)
            dbg = ts.cpu().numpy().view(np.uint32).reshape(-1)[(1 << 22) + 4 * 283:(1 << 22) + 4 * 283 + 4]
            clk = "; exits: %d exact node tests, %d triangle candidates, %.0f %% of the cycles between exit and re-entry, %d walks" % (dbg[0], dbg[1], 100.0 * (int(dbg[2]) << 4) / raw[283, 2], dbg[3])
            clk += "; tile 283 ran at %.0f MHz, %.0f %% of its cycles inside %d walks" % (raw[283, 2] / dur[283], 100.0 * ((raw[283, 3] & 0xffffff) << 4) / raw[283, 2], raw[283, 3] >> 24)
        if os.environ.get("RT_EXP_STAGES"):   # (build with -DRT_STAGE_COUNT and a block generated with RT_GEN_COUNT=1)
            for tile in (283, 284, 100):
                stg = ts.cpu().numpy().view(np.uint32).reshape(-1)[(1 << 22) + (1 << 20) + 4 * tile:(1 << 22) + (1 << 20) + 4 * tile + 4]
                clk += "; tile %d: %d node and %d triangle iterations, %d went on to stage 2, %d to stage 3" % (tile, stg[3], stg[0], stg[1], stg[2])
        print("%-24s %-15s kernel %.3f ms; tile 283: %4.0f us, tile 284: %4.0f us, tile 271: %4.0f us; mean %.1f; rays traced %d%s" % (
            what, fl, st["kernel_ms"], dur[283], dur[284], dur[271], dur.mean(), st["rays_traced"], clk))
