#!/usr/bin/env python3
"""Wave-level iteration counts of the instrumented kernel on the headline frame (how often each code path of the
traversal runs); run from the repository root."""
import sys
sys.path.insert(0,'.')
import raytracer_amd as rt
ds = rt.DeviceScene(rt.Scene("scenes/meshDragon.json", 3840, 2160, 3))
rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_COUNT_TESTS)
names=["traces","node_it","node_exact","node_tie","tri_it","tri_s2","tri_s3","tri_fin"]
print({n: v for n, v in zip(names, st["wave_paths"])}, st["node_tests"], st["tri_tests"], st["rays"])
