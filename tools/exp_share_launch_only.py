#!/usr/bin/env python3
"""What one GPU of N renders per frame when nothing but the render launches is in the loop: the bands of rank 0 of N on K
streams in turn, no exchange, no host work besides the launch (bench.py's rehearsal adds torch's gather with a world of one).
   python tools/exp_share_launch_only.py"""
import sys
import time
sys.path.insert(0, '.')
import torch
import raytracer_amd as rt

w, h = 3840, 2160
ds0 = rt.DeviceScene(rt.Scene("scenes/meshDragon.json", w, h, 3))
import os
KS = tuple(int(x) for x in os.environ.get('RT_EXP_STREAMS', '1,2,3,4').split(','))
for n in (8, 4, 2):
    nb = ds0.band_count(16, 0, n)
    for k in KS:
        streams = [torch.cuda.Stream() for _ in range(k)]
        ctx = [ds0] + [ds0.fork() for _ in range(k - 1)]
        bufs = [torch.zeros((nb * 16, w, 3), dtype=torch.uint8, device="cuda") for _ in range(k)]
        def frames(count):
            for f in range(count):
                s = f % k
                ctx[s].render_bands_into(bufs[s].data_ptr(), 16, 0, n, stream=streams[s].cuda_stream, flags=rt.RT_RENDER_STRIPS if k > 1 else 0)
        frames(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames(200)
        t_launch = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("share of %d, %d stream(s): %.4f ms per frame (host time to launch a frame %.4f ms); ideal %.4f" % (n, k, dt * 1e3 / 200, t_launch * 1e3 / 200, 1.18 / n))
