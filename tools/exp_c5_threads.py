"""Measurement helper: K frames of one scene rendered by T host threads, each with a context and a stream of its own
(what overlapping independent frames - or independent parts of one frame - on one GPU gains).  Prints ms per frame."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt

scene_file = sys.argv[1] if len(sys.argv) > 1 else "scenes/meshTransparentBunny.json"
W, H = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "3840x2160").split("x"))
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sc = rt.Scene(rt.asset(scene_file), W, H, 3)
root = rt.DeviceScene(sc, 0)
for T in (1, 2, 3):
    forks = [root] + [root.fork() for _ in range(T - 1)]
    streams = [torch.cuda.Stream() for _ in range(T)]
    frames = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(T)]

    def work(k, n):
        for _ in range(n):
            forks[k].render_into(frames[k].data_ptr(), stream=streams[k].cuda_stream)

    for k in range(T): work(k, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k, K)) for k in range(T)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(frames[0], f) for f in frames[1:])
    print(f"{scene_file} {W}x{H} threads {T}: {1e3 * dt / (T * K):.3f} ms per frame, frames equal {same}", flush=True)
