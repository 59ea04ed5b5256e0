#!/usr/bin/env python3
"""How long do the longest waves of the headline frame live when they have the chip (nearly) to themselves, and how does
that change with what else is resident?  Diagnostic build (tools/build_variant.sh times -DRT_WAVE_TIMES, selected with RT_AMD_LIB).
   RT_AMD_LIB=build/variants/librt_amd_times.so python tools/exp_lone_wave.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import raytracer_amd as rt

w, h = 3840, 2160
ds = rt.DeviceScene(rt.Scene("scenes/meshDragon.json", w, h, 3))
tx = w // 8
rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
ts = torch.zeros((h, w), dtype=torch.float32, device="cuda")


def lives(n_tiles):
    raw = ts.cpu().numpy().view(np.uint32).reshape(-1)[:4 * n_tiles].reshape(n_tiles, 4)
    start = raw[:, 0].astype(np.int64); end = raw[:, 1].astype(np.int64)
    end = np.where(end < start, end + (1 << 32), end)
    return (end - start) / 100.0, start - start.min()


for flags, what in ((0, "default"), (rt.RT_RENDER_EVERY_TRIANGLE, "every triangle")):
    for ty in (103, 81):
        for _ in range(2):
            ts.zero_()
            st = ds.render_into(rgb.data_ptr(), ty * 8, ty * 8 + 8, t_ptr=ts.data_ptr(), flags=rt.RT_RENDER_TIME | flags, want_stats=True)
        dur, _ = lives(tx)
        order = np.argsort(-dur)[:6]
        print("%s, tile row %d alone: kernel %.3f ms; wave life us mean %.1f max %.1f; longest: %s" % (
            what, ty, st["kernel_ms"], dur.mean(), dur.max(), "  ".join("(%d) %.0f" % (i, dur[i]) for i in order)))
# growing windows of rows around tile row 103: more and more waves resident beside the long ones
for half in (0, 1, 2, 4, 8, 16, 32, 64, 103):
    r0, r1 = (103 - half) * 8, (103 + half + 1) * 8
    for _ in range(2):
        ts.zero_()
        st = ds.render_into(rgb.data_ptr(), r0, min(r1, h), t_ptr=ts.data_ptr(), flags=rt.RT_RENDER_TIME, want_stats=True)
    n_rows = (min(r1, h) - r0) // 8
    dur, start = lives(tx * n_rows)
    i = half * tx + 283   # tile (283, 103) inside the window
    print("tile rows %d..%d (%d waves): kernel %.3f ms; tile (283,103) lives %.0f us from %.3f ms; longest %.0f; mean %.1f" % (
        103 - half, 103 + half, tx * n_rows, st["kernel_ms"], dur[i], start[i] / 1e5, dur.max(), dur.mean()))
