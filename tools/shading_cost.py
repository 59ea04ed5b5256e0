#!/usr/bin/env python3
"""Splits the kernel time of a mesh scene into walk and shading by rendering variants of the scene (one GPU):
   the scene as it is; without shadow rays (same shading arithmetic, primary walks only); with the ambient light only
   (primary walks, next to no shading).
   python tools/shading_cost.py [--scene meshDragon] [--size 3840x2160]"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import raytracer_amd as rt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="meshDragon")
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    assets = os.path.abspath(os.path.join(ROOT, "assets"))
    base = tempfile.mkdtemp(prefix="rt_shading_cost_")
    os.symlink(os.path.join(assets, "meshes"), os.path.join(base, "meshes"))
    os.symlink(os.path.join(assets, "resources.json"), os.path.join(base, "resources.json"))
    os.mkdir(os.path.join(base, "scenes"))
    with open(os.path.join(assets, "scenes", a.scene + ".json")) as f:
        scene = json.load(f)
    variants = {"as it is": scene}
    v = json.loads(json.dumps(scene)); v.setdefault("params", {})["enable_shadows"] = "false"
    variants["no shadow rays"] = v
    v = json.loads(json.dumps(scene)); v["lights"] = [l for l in v["lights"] if l["__type__"] == "ambient_light"]
    variants["ambient light only"] = v
    for kind in ("point_light", "directional_light", "spot_light"):
        if any(l["__type__"] == kind for l in scene["lights"]):
            v = json.loads(json.dumps(scene)); v["lights"] = [l for l in v["lights"] if l["__type__"] in (kind, "ambient_light")]
            variants["ambient + %s only" % kind] = v
    print("%-34s %12s %12s %10s" % ("variant", "rays", "traced", "kernel_ms"))
    for name, sc in variants.items():
        path = os.path.join(base, "scenes", "v.json")
        with open(path, "w") as f:
            json.dump(sc, f)
        ds = rt.DeviceScene(rt.Scene(path, w, h, 3, base_dir=base))
        best = None
        for _ in range(a.reps):
            _, _, _, st = ds.render_host(want_ids=False, flags=rt.RT_RENDER_TIME)
            best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
        print("%-34s %12d %12d %10.3f" % (name, best["rays"], best["rays_traced"], best["kernel_ms"]), flush=True)
        ds.close()


if __name__ == "__main__":
    main()
