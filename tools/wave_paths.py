#!/usr/bin/env python3
"""Wave-level iteration counts on the headline frame (how often each code path of the traversal runs); run from the
repository root with a build that keeps them: tools/build_variant.sh paths -DRT_PATH_COUNTS [-DRT_PATH_GENERIC], then
   RT_AMD_LIB=build/variants/librt_amd_paths.so python tools/wave_paths.py [--instrumented]
--instrumented: the counts of the instrumented variant (full shadow walks, any build).  With -DRT_PATH_GENERIC the fourth
count is the number of node iterations outside the scan loops instead of the number of ties."""
import sys
sys.path.insert(0, '.')
import raytracer_amd as rt
ds = rt.DeviceScene(rt.Scene("scenes/meshDragon.json", 3840, 2160, 3))
rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_COUNT_TESTS if "--instrumented" in sys.argv else 0)
names = ["traces", "node_it", "node_exact", "node_tie", "tri_it", "tri_s2", "tri_s3", "tri_fin"]
print({n: v for n, v in zip(names, st["wave_paths"])}, st["node_tests"], st["tri_tests"], st["rays"])
