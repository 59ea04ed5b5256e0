#!/usr/bin/env python3
"""What the steps of the assembly walk cost a wave, fitted from the waves themselves: tile rows of the 4K meshDragon frame rendered
one at a time (8x8 tiles), every wave's life in shader clock cycles against what it counted (diagnostic build:
tools/build_variant.sh count -DRT_WAVE_TIMES -DRT_WAVE_TIMES_CLK -DRT_STAGE_COUNT with a block generated with RT_GEN_COUNT=1).
   RT_AMD_TILE=8 RT_AMD_LIB=build/variants/librt_amd_count.so python tools/exp_walk_cost_fit.py [out.npz]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from scipy.optimize import nnls
import raytracer_amd as rt

w, h = 3840, 2160
tx = w // 8
ds = rt.DeviceScene(rt.Scene("scenes/meshDragon.json", w, h, 3))
rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
ts = torch.zeros((h, w), dtype=torch.float32, device="cuda")
rows = []
for ty in range(40, 200, 3):
    for _ in range(2):
        ts.zero_()
        ds.render_into(rgb.data_ptr(), ty * 8, ty * 8 + 8, t_ptr=ts.data_ptr(), flags=rt.RT_RENDER_TIME, want_stats=True)
    flat = ts.cpu().numpy().view(np.uint32).reshape(-1)
    raw = flat[:4 * tx].reshape(tx, 4)
    dbg = flat[(1 << 22):(1 << 22) + 4 * tx].reshape(tx, 4)
    stg = flat[(1 << 22) + (1 << 20):(1 << 22) + (1 << 20) + 8 * tx].reshape(tx, 8)
    for i in range(tx):
        # life, cycles in walks, cycles outside the block within walks | triangle iterations, stage 2, stage 3, node iterations, blocks entered, leaves, margin tests, pops | exact exits, candidate exits, walks
        rows.append([raw[i, 2], (raw[i, 3] & 0xffffff) << 4, int(dbg[i, 2]) << 4] + [int(x) for x in stg[i]] + [int(dbg[i, 0]), int(dbg[i, 1]), int(dbg[i, 3])])
a = np.array(rows, dtype=np.float64)
if len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], a=a)
names = ["triangle iteration", "  went on to stage 2", "  went on to stage 3", "node iteration", "block of children entered", "leaf entered", "margin test of a child", "pop",
         "exit for an exact node test", "exit for triangle candidates", "walk"]
inside = a[:, 1] - a[:, 2]                # cycles inside the block
X = a[:, 3:14]
keep = a[:, 13] > 0
print("%d waves with a walk; cycles inside the block: mean %.0f, longest %.0f" % (keep.sum(), inside[keep].mean(), inside[keep].max()))
for what, y in (("inside the block", inside), ("whole life", a[:, 0])):
    Xk = np.hstack([X[keep], np.ones((keep.sum(), 1))])
    coef, _ = nnls(Xk, y[keep])
    pred = Xk @ coef
    print("fit of the cycles %s (non-negative least squares, R^2 %.4f):" % (what, 1 - ((y[keep] - pred) ** 2).sum() / ((y[keep] - y[keep].mean()) ** 2).sum()))
    tot = Xk.sum(axis=0) * coef
    for n, c, t in zip(names + ["per wave"], coef, tot):
        print("   %-30s %8.1f cycles each, %5.1f %% of all cycles" % (n, c, 100 * t / y[keep].sum()))
    big = np.argsort(-y[keep])[:5]
    for b in big:
        print("   a long wave: %.0f cycles measured, %.0f fitted; counts %s" % (y[keep][b], pred[b], " ".join("%d" % v for v in X[keep][b])))
