#!/usr/bin/env python3
"""Wave-level iteration counts of the instrumented kernel variant for any scene (full shadow walks; compiled walk):
   python tools/wave_paths_scene.py SCENE WxH   -> traces, node / triangle iterations of the waves, per-lane tests, lanes per iteration."""
import sys
sys.path.insert(0, '.')
import raytracer_amd as rt
name, size = sys.argv[1], sys.argv[2]
w, h = (int(x) for x in size.split("x"))
ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % name, w, h, 3))
rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_COUNT_TESTS)
names = ["traces", "node_it", "node_exact", "node_tie", "tri_it", "tri_s2", "tri_s3", "tri_fin"]
p = {n: v for n, v in zip(names, st["wave_paths"])}
print(name, size, p, "node_tests", st["node_tests"], "tri_tests", st["tri_tests"], "leaf", st.get("leaf_visits"), "rays", st["rays"], "queued", st.get("queued"))
print("lanes per node iteration %.1f, per triangle iteration %.1f, triangle iterations per trace %.0f, node iterations per trace %.0f" %
      (st["node_tests"] / max(p["node_it"], 1), st["tri_tests"] / max(p["tri_it"], 1), p["tri_it"] / max(p["traces"], 1), p["node_it"] / max(p["traces"], 1)))
