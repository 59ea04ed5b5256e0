#!/usr/bin/env python3
"""Kernel time and Mrays/s of the render path on several scenes and sizes (one GPU): a table for the design notes.
   python tools/bench_scenes.py [--sizes 512x512,3840x2160] [scene ...]"""
import argparse
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import time
import raytracer_amd as rt

DEFAULT = ["sphere", "lambertBPHSpheres", "heliumAttenuatedSphere", "boxedSpheres", "meshTorus", "meshReflectedSquircle",
           "meshMedBunny", "meshTransparentBunny", "meshDragon"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="512x512,3840x2160")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("scenes", nargs="*", default=DEFAULT)
    a = ap.parse_args()
    print("%-26s %11s %12s %12s %10s %10s %10s" % ("scene", "size", "rays", "traced", "kernel_ms", "Mrays/s", "host_ms"))
    for name in a.scenes:
        for size in a.sizes.split(","):
            w, h = (int(x) for x in size.split("x"))
            ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
            best, host_ms = None, 1e30
            for _ in range(a.reps):
                t0 = time.perf_counter()
                _, _, _, st = ds.render_host(want_ids=False, flags=rt.RT_RENDER_TIME)
                host_ms = min(host_ms, (time.perf_counter() - t0) * 1e3)   # device buffers, kernel, copy of the frame to the host
                best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
            print("%-26s %11s %12d %12d %10.3f %10.1f %10.3f" % (name, size, best["rays"], best["rays_traced"], best["kernel_ms"],
                                                              best["rays"] / best["kernel_ms"] / 1e3, host_ms), flush=True)
            ds.close()


if __name__ == "__main__":
    main()
