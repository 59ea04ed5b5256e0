#!/usr/bin/env python3
"""Kernel time of every rank's share of the headline frame for N = 1, 2, 4, 8 ranks, all measured on ONE GPU (what the
band partition alone allows in strong scaling, before any exchange): python tools/band_share_times.py [--band-rows 16]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracer_amd as rt

ap = argparse.ArgumentParser()
ap.add_argument("--band-rows", type=int, default=16)
ap.add_argument("--scene", default="scenes/meshDragon.json")
a = ap.parse_args()
W, H = 3840, 2160
ds = rt.DeviceScene(rt.Scene(a.scene, W, H, 3))
buf = torch.zeros((H + 64, W, 3), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream()
for n in (1, 2, 4, 8):
    times = []
    for r in range(n):
        best = 1e9
        for _ in range(4):
            st = ds.render_bands_into(buf.data_ptr(), a.band_rows, r, n, stream=s.cuda_stream, flags=rt.RT_RENDER_TIME, want_stats=True)
            best = min(best, st["kernel_ms"])
        times.append(best)
    print("N=%d  band rows %d: per-rank kernel ms %s  max %.3f  ideal %.3f  efficiency %.2f" %
          (n, a.band_rows, " ".join("%.3f" % t for t in times), max(times), times[0] / n if n == 1 else base / n, (base / n) / max(times) if n > 1 else 1.0), flush=True)
    if n == 1:
        base = times[0]
