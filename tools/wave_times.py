#!/usr/bin/env python3
"""Life spans of the waves of one frame (diagnostic build: tools/build_variant.sh times -DRT_WAVE_TIMES, selected with
RT_AMD_LIB): how many waves are resident over time, how long waves live, how evenly the XCDs finish.
   RT_AMD_LIB=build/variants/librt_amd_times.so python tools/wave_times.py [scene] [WxH]   (from the repository root)"""
import sys
sys.path.insert(0, '.')
import numpy as np
import raytracer_amd as rt

scene = sys.argv[1] if len(sys.argv) > 1 else "meshDragon"
w, h = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3840x2160").split("x"))
ds = rt.DeviceScene(rt.Scene("scenes/%s.json" % scene, w, h, 3))
counting = len(sys.argv) > 3 and sys.argv[3] == "count"   # the counting variant: the queue waves also log their test counts
for _ in range(2):
    rgb, ids, ts, st = ds.render_host(flags=rt.RT_RENDER_TIME | (rt.RT_RENDER_COUNT_TESTS if counting else 0))
n = ((w + 7) // 8) * ((h + 7) // 8)
raw = ts.view(np.uint32).reshape(-1)[:4 * n].reshape(n, 4)
start = raw[:, 0].astype(np.int64); end = raw[:, 1].astype(np.int64)
end = np.where(end < start, end + (1 << 32), end)
t0 = start.min(); start -= t0; end -= t0
total = end.max()
print("kernel_ms %.3f; first wave start to last wave end: %.3f ms (100 MHz clock); %d waves" % (st["kernel_ms"], total / 1e5, n))
dur = (end - start) / 100.0
print("wave life (us): mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
# resident waves over time, in 20 slices
edges = np.linspace(0, total, 21)
print("slice  resident waves (avg)   waves started   waves ended")
for k in range(20):
    a, b = edges[k], edges[k + 1]
    overlap = np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / (b - a)
    print("%5d  %12.0f  %14d  %12d" % (k, overlap, ((start >= a) & (start < b)).sum(), ((end > a) & (end <= b)).sum()))
xcc = raw[:, 3] & 0xf
print("per XCD: waves, busy wave-ms, last end (ms)")
for x in np.unique(xcc):
    m = xcc == x
    print("  xcd %d: %7d  %8.2f  %6.3f" % (x, m.sum(), dur[m].sum() / 1e3, end[m].max() / 1e5))
hw = raw[:, 2]
slot = (xcc.astype(np.int64) << 16) | (hw & 0x7fff)
cu = (xcc.astype(np.int64) << 16) | (hw & 0x7f00)
simd = (xcc.astype(np.int64) << 16) | (hw & 0x7f30)
print("distinct wave slots seen %d, SIMDs %d, compute units %d" % (len(np.unique(slot)), len(np.unique(simd)), len(np.unique(cu))))
u_simd, inv = np.unique(simd, return_inverse=True)
slots_per_simd = np.array([len(np.unique(slot[inv == k])) for k in range(len(u_simd))])
print("wave slots used per SIMD: min %d max %d mean %.2f; histogram %s" % (slots_per_simd.min(), slots_per_simd.max(), slots_per_simd.mean(), np.bincount(slots_per_simd).tolist()))
tx = (w + 7) // 8
order = np.argsort(-dur)[:24]
print("longest waves: (tile x, tile y) life us, start ms")
print("  " + "  ".join("(%d,%d) %.0f @%.2f" % (i % tx, i // tx, dur[i], start[i] / 1e5) for i in order))
ty = (h + 7) // 8
grid = dur.reshape(ty, tx)
print("mean life by tile column block (16 blocks):", [int(grid[:, k * tx // 16:(k + 1) * tx // 16].mean()) for k in range(16)])
print("centre columns %d..%d mean life: %s" % (tx // 2 - 3, tx // 2 + 2, [int(grid[:, c].mean()) for c in range(tx // 2 - 3, tx // 2 + 3)]))
# ---- the waves of the queue launches of a frame rendered from queues (logged behind the tiles' records) ----
words = ts.view(np.uint32).reshape(-1)
if 4 * n + 4 < words.size and st.get("from_queues"):
    logged = int(words[4 * n])
    room = (words.size - 4 * n - 4) // 4
    m = min(logged, room)
    q = words[4 * n + 4:4 * n + 4 + 4 * m].reshape(m, 4)
    qs = q[:, 0].astype(np.int64); qe = q[:, 1].astype(np.int64)
    qe = np.where(qe < qs, qe + (1 << 32), qe)
    print("queue launches: %d waves logged (%d ended)" % (m, logged))
    tests_sum = q[:, 2] >> 2
    q[:, 2] &= 3
    for gen in np.unique(q[:, 2]):
        g = q[:, 2] == gen
        if counting:
            life_g = (qe[g] - qs[g]) / 100.0
            top = np.argsort(-life_g)[:12]
            print(" generation %d, longest waves: life us / node + triangle tests of the wave / of its busiest lane" % gen)
            print("   " + "  ".join("%.0f/%d/%d" % (life_g[i], tests_sum[g][i], q[g][i, 3]) for i in top))
            print("   all waves: tests per wave mean %.0f p99 %.0f max %.0f; busiest lane mean %.0f p99 %.0f max %d; correlation of life with the wave's tests %.3f, with the busiest lane's %.3f"
                  % (tests_sum[g].mean(), np.percentile(tests_sum[g], 99), tests_sum[g].max(), q[g][:, 3].mean(), np.percentile(q[g][:, 3], 99), q[g][:, 3].max(),
                     np.corrcoef(life_g, tests_sum[g])[0, 1], np.corrcoef(life_g, q[g][:, 3])[0, 1]))
            continue
        a, b = qs[g] - t0, qe[g] - t0
        life = (b - a) / 100.0
        span = max(b.max() - a.min(), 1) / 1e5
        print(" generation %d: %d waves, launch %.3f ms (from %.3f ms); life us mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f; busy wave-ms %.1f = %.0f waves on average"
              % (gen, g.sum(), span, a.min() / 1e5, life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), life.sum() / 1e3, life.sum() / 1e3 / span))
        edges2 = np.linspace(a.min(), max(b.max(), a.min() + 10), 11)
        res = [int(np.clip(np.minimum(b, edges2[k + 1]) - np.maximum(a, edges2[k]), 0, None).sum() / (edges2[k + 1] - edges2[k])) for k in range(10)]
        print("   resident waves by tenth of the launch:", res)
        longest = np.argsort(-life)[:8]
        print("   longest: " + "  ".join("chunk %d %.0f us @%.2f" % (q[g][i, 3], life[i], a[i] / 1e5) for i in longest))
