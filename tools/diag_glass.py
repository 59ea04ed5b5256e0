import sys, os
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt
w, h = 1920, 1080
ds = rt.DeviceScene(rt.Scene("scenes/meshTransparentBunny.json", w, h, 3))
a, ida, ta, sa = ds.render_host(flags=rt.RT_RENDER_QUEUED)
b, idb, tb, sb = ds.render_host(flags=rt.RT_RENDER_STACKED)
print("queued vs stacked: rgb pixels differing", int((a != b).any(axis=2).sum()), "ids differing", int((ida != idb).sum()), "t differing", int((ta.view(np.uint32) != tb.view(np.uint32)).sum()))
ys, xs = np.nonzero((a != b).any(axis=2))
print("where:", list(zip(xs[:10].tolist(), ys[:10].tolist())))
for k in ("rays", "shadow_rays", "rays_traced"): print(k, sa[k], sb[k])
np.save("gpurun_out/glass_q.npy", a); np.save("gpurun_out/glass_s.npy", b)
