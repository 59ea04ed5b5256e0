#!/usr/bin/env python3
"""Headline benchmark: Mrays/s on scenes/meshDragon.json at 3840x2160 (max depth 3), N GPUs of one node.

One step = one full frame.  With N > 1 the frame is cut into 16-row bands dealt round-robin to the
ranks (one process per GPU); every rank renders its bands packed and a single RCCL gather collects
the framebuffer tiles on rank 0 (total work fixed: strong scaling).  Inputs are resident in HBM
before the timed region; the output stays on the device.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  `rays` are calls of the closest-hit routine the reference performs for
the frame (primary + shadow + bounce); the kernels report the same count from their own counters.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BAND_ROWS = 16
PEAK_FP32_TFLOPS = 157.3     # MI355X vector FP32 peak (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0

# Algorithmic work per test of the minimal equivalent formulation (slab node test, the reference's
# Cramer triangle solve, sphere quadratic, shading per ray): FLOP.
F_NODE, F_TRI, F_SPH, F_SHADE = 24, 61, 20, 150
# Algorithmic bytes: node record, triangle reference + positions, hit normals + ray record.
B_NODE, B_TRI, B_RAY = 32, 40, 136


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="scenes/meshDragon.json",
                    help="scene file under assets/, or icosphere:K for the synthetic mesh (20*4**K triangles, K=6 or 7)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-band-stride", type=int, default=1, help="oracle renders every k-th 16-row band (1 = the whole frame)")
    return ap.parse_args()


def committed_counters(source_sha256, workload=None):
    """The rocprofv3 counter passes committed under profiles/ (r*_counters.json, tools/profile_to_json.py), if they were
    taken with exactly the kernel sources that are running now (digest of the source files recorded in the profile);
    hardware counters cannot be read from inside the benchmark, and numbers of other code are not quoted: None then."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")), reverse=True):
        try:
            with open(path) as f:
                j = json.load(f)
        except (OSError, ValueError):
            continue
        if j.get("source_sha256") == source_sha256 and (workload is None or j.get("workload") == workload):
            j["file"] = os.path.relpath(path, ROOT)
            return j
    return None


def cpu_baseline(args, threads):
    """Time the CPU oracle on every k-th 16-row band of the same frame (a bounded, representative sample)."""
    from oracle_api import OracleScene
    o = OracleScene(args.scene, args.width, args.height, args.depth, base_dir=args.base_dir)
    import numpy as np
    import ctypes as C
    from oracle_api import Counters, lib
    rgb = np.zeros((args.height, args.width, 3), np.uint8)
    rays = 0
    rows = 0
    t0 = time.perf_counter()
    for b in range(0, (args.height + BAND_ROWS - 1) // BAND_ROWS, args.cpu_band_stride):
        ct = Counters()
        r0, r1 = b * BAND_ROWS, min((b + 1) * BAND_ROWS, args.height)
        lib().oracle_render_rows(o.h, r0, r1, rgb.ctypes.data, None, None, C.byref(ct), threads)
        rays += ct.rays
        rows += r1 - r0
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": ("the whole %dx%d frame: %d rows, %d rays in %.2f s" % (args.width, args.height, rows, rays, dt))
                      if args.cpu_band_stride == 1 else
                      ("every %d-th 16-row band of the same %dx%d frame: %d rows, %d rays in %.2f s" %
                       (args.cpu_band_stride, args.width, args.height, rows, rays, dt))}


def main():
    args = parse_args()
    # Rank 0 prints exactly one line on stdout.  Libraries (RCCL's version banner among them) write to file descriptor 1
    # behind Python's back, so descriptor 1 points at stderr until the result line is due.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # A rank of six or more GPUs keeps four short frames in flight (below); with the collective's stream that is more streams than
    # the four hardware queues a process gets by default, and streams that share a queue run one after the other.  Eight queues
    # for such a process (measured on one GPU with the bands of one rank of 8: 0.223 -> 0.201 ms per frame); it has to be in the
    # environment before the HIP runtime starts.
    parts = max(int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RT_BENCH_SHARE_OF", "0") or 0))
    if parts >= 6 and int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8 and os.environ.get("RT_BENCH_KEEP_QUEUES") != "1":
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import torch
    import torch.distributed as dist
    import raytracer_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    W, H = args.width, args.height
    base_dir = None
    if args.scene.startswith("icosphere:"):
        import tempfile
        from raytracer_amd.synthetic import write_icosphere_scene
        base_dir = tempfile.mkdtemp(prefix="rt_icosphere_rank%d_" % rank)
        args.scene, _ = write_icosphere_scene(base_dir, int(args.scene.split(":")[1]))
    args.base_dir = base_dir
    scene = rt.Scene(args.scene, W, H, args.depth, base_dir=base_dir)
    info = scene.info()
    ds = rt.DeviceScene(scene, device=local_rank)
    # Rehearsals on one GPU: RT_BENCH_FORCE_DIST=1 runs the exchange with a world of one; RT_BENCH_SHARE_OF=N renders only
    # the bands rank 0 of N would own (what one GPU of N does per frame).
    share_of = int(os.environ.get("RT_BENCH_SHARE_OF", "0")) if world == 1 else 0
    part_rank, part_world = (0, share_of) if share_of > 1 else (rank, world)
    nb_max = ds.band_count(BAND_ROWS, 0, part_world)       # rank 0 owns the most bands
    dist_on = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1" or share_of > 1
    if dist_on and world == 1:   # a rehearsal in a single process: it is its own rendezvous
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(key, val)
    if dist_on and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=dev)
    # Two or three frame tiles per rank: the gather of frame k runs on the collective's stream while frame k+1 is rendered.
    two_deep = dist_on and os.environ.get("RT_BENCH_ONE_STREAM") != "1"
    # (A rank's launch lasts as long as its longest wave, a silhouette tile's 0.5 ms, however small its share: what a rank
    #  renders per unit of time is the number of frames in flight over that.  Measured on one GPU with the bands of one rank
    #  (tools/exp_in_flight.sh, tools/exp_share_launch_only.py): of 8, 0.31 / 0.22 / 0.20 ms per frame with two / three / four in
    #  flight (four need the eight hardware queues above; the launches alone, without an exchange, take 0.28 / 0.21 / 0.17); of 4,
    #  0.36 / 0.335 with two / three, no gain beyond; of 2, two frames already reach the single-GPU rate.)
    in_flight = max(2, int(os.environ.get("RT_BENCH_IN_FLIGHT", "4" if part_world >= 6 else ("3" if part_world >= 4 else "2")))) if two_deep else 1
    local = [torch.zeros((nb_max * BAND_ROWS, W, 3), dtype=torch.uint8, device=dev) for _ in range(max(in_flight, 2) if dist_on else 1)]
    gathered = [[torch.zeros_like(local[0]) for _ in range(world)] for _ in local] if (dist_on and rank == 0) else None
    pending = [None] * len(local)
    # With several GPUs a rank's share of a frame is short, and it ends with the few long waves of its silhouette tiles
    # while most of the chip idles.  So frames are pipelined two (or three) deep there: frame k+1 is rendered on a second stream
    # into the second tile while frame k finishes and is gathered.  One GPU: one stream, launches back to back.
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(in_flight - 1)]
    stream = streams[0]
    # one render context per stream on the one uploaded scene (renders of one rt_dscene must not overlap: it owns one set
    # of counters and queues)
    scenes = [ds] + [ds.fork() for _ in range(in_flight - 1)]
    # frames in flight fill the chip between them: the tiles of every launch go to the dies in strips whatever their number
    pipeline_flags = rt.RT_RENDER_STRIPS if two_deep else 0

    def step(k, flags=0, want_stats=False, events=None, n_slots=None):
        b = k % (n_slots or len(local))
        sk = streams[b % len(streams)]
        with torch.cuda.stream(sk):
            if pending[b] is not None:
                pending[b].wait()          # the render stream waits for the gather that last read this tile
                pending[b] = None
            if events:
                events[0].record(sk)
            st = scenes[b % len(scenes)].render_bands_into(local[b].data_ptr(), BAND_ROWS, part_rank, part_world, stream=sk.cuda_stream,
                                                           flags=flags | pipeline_flags, want_stats=want_stats)
            if events:
                events[1].record(sk)
            if dist_on:
                pending[b] = dist.gather(local[b], gathered[b] if rank == 0 else None, dst=0, async_op=True)
        return st

    def drain():
        for b in range(len(pending)):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    # ray counts of this rank's share (deterministic; read back once, outside the timed region)
    st = step(0, want_stats=True)
    drain()
    rays_t = torch.tensor([st["rays"], st["shadow_rays"]], dtype=torch.int64, device=dev)
    if dist_on:
        dist.all_reduce(rays_t)
    rays_total, shadow_total = int(rays_t[0]), int(rays_t[1])

    def timed(steps, one_at_a_time):
        """`steps` frames between barriers.  one_at_a_time: a frame is only launched when the one before it has landed on
        rank 0 (its window is first launch -> last tile in place); otherwise frames are kept in flight as configured.
        Returns (seconds, mean HIP-event duration of the render launches in ms), both the maximum over the ranks."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k, events=ev[k], n_slots=1 if one_at_a_time else None)
            if one_at_a_time and dist_on:
                drain()
                torch.cuda.synchronize()
        drain()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(steps, 1)
        tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        if dist_on:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0]), float(tt[1])

    for k in range(args.warmup):
        step(k)
    drain()
    # THE timed region: exactly `steps` frames.  One GPU: launches back to back on one stream.  Several GPUs: frames in flight.
    elapsed, kernel_ms = timed(args.steps, one_at_a_time=False)
    # Several GPUs: the same number of frames again, one at a time (no pipelining), so that the line carries both the
    # throughput with frames in flight (`value`) and the time of a single frame from its first launch to its last tile on rank 0.
    single = timed(args.steps, one_at_a_time=True) if dist_on else None

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        mrays = rays_total / (elapsed / args.steps) / 1e6
        # roofline of the dominant (only) kernel, from the reference's test counts for this exact frame
        ref = None
        try:
            with open(os.path.join(ROOT, "tests", "golden", "reference_counters.json")) as f:
                g = json.load(f)
            key = "meshDragon_%dx%d" % (W, H)
            if args.scene.endswith("meshDragon.json") and args.depth == 3 and key in g:
                ref = g[key]
        except OSError:
            pass
        counts_from = "reference run (tests/golden/reference_counters.json)"
        if ref is None and world == 1:
            # no recorded reference counts for this workload: take them from the instrumented kernel variant, whose
            # counts the parity tests hold identical to the oracle's (one extra launch, outside the timed region)
            sc = ds.render_bands_into(local[0].data_ptr(), BAND_ROWS, part_rank, part_world, stream=stream.cuda_stream,
                                      flags=rt.RT_RENDER_COUNT_TESTS, want_stats=True)
            ref = {"rays": sc["rays"], "node_tests": sc["node_tests"], "tri_tests": sc["tri_tests"], "sphere_tests": sc["sphere_tests"]}
            counts_from = "instrumented kernel variant"
        roofline = None
        if ref and ref["rays"] == rays_total:
            flop = ref["node_tests"] * F_NODE + ref["tri_tests"] * F_TRI + ref.get("sphere_tests", 0) * F_SPH + ref["rays"] * F_SHADE
            byts = ref["node_tests"] * B_NODE + ref["tri_tests"] * B_TRI + ref["rays"] * B_RAY
            # per launch: this rank's share of the frame over the launch's own duration; with frames pipelined two deep
            # two launches share the GPU and their durations overlap, so the rate is taken over the time per frame then
            t_ms = ms_per_step if two_deep else kernel_ms
            ach = flop / world / (t_ms * 1e-3) / 1e12
            # The path is neither a dense contraction nor memory-bound: the roof that binds is FP32 vector issue.  The HBM
            # view is reported beside it (hbm_*): measured bytes per launch over the same kernel time, against 8 TB/s.
            # (counter passes are committed per workload: the headline frame, and the other configurations at 3840x2160)
            workload = "%s %dx%d depth %d" % (os.path.join("scenes", os.path.basename(args.scene)), W, H, args.depth)
            prof = committed_counters(rt.kernel_source_sha256(), workload) if (world == 1 and not share_of) else None
            roofline = {"bound": "valu_fp32", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_FP32_TFLOPS,
                        # (the kernel proves most triangle tests of a leaf rejected for a whole wave without executing them, and stops
                        #  shadow rays at their first hit: `achieved` prices the tests the reference makes, executed_* what ran)
                        "frac_is": "FLOP of the node / triangle / sphere tests and shading the REFERENCE performs for this frame, over the kernel's time and the FP32 peak: work delivered, not work executed (see executed_tflops)",
                        # HBM bytes per launch (FETCH_SIZE + WRITE_SIZE of the counter passes committed for exactly these kernel sources; null otherwise)
                        "traffic": prof.get("traffic_bytes_per_launch") if prof else None,
                        "kernel": "render_wave", "kernel_ms": kernel_ms, "launches_overlap": bool(two_deep), "test_counts_from": counts_from, "flop_per_ray": flop / ref["rays"],
                        # bytes of node / triangle records the reference's tests would touch, per second: served by the scalar cache and L2
                        # (a record is fetched once per wave, not per ray), NOT memory traffic - the HBM view is hbm_*
                        "record_bytes_cache_level_GBps": byts / world / (t_ms * 1e-3) / 1e9, "hbm_peak_GBps": PEAK_HBM_GBS}
            if prof:
                # what the vector units executed, beside the algorithmic figure: FP32 add + mul + 2 x fma + transcendental wave-instructions x 64 lanes
                c = prof.get("counters_per_launch", {})
                roofline["counters_from"] = prof["file"]
                roofline["executed_tflops"] = prof["executed_fp32_flop_per_launch"] / (kernel_ms * 1e-3) / 1e12
                roofline["executed_frac"] = roofline["executed_tflops"] / PEAK_FP32_TFLOPS
                cycles = kernel_ms * 1e-3 * 2.4e9 * 1024      # SIMD-cycles of the launch at the 2.4 GHz peak clock
                roofline["executed_frac_of_fma_free_peak"] = roofline["executed_tflops"] / (PEAK_FP32_TFLOPS / 2)   # the reference arithmetic has no fused multiply-adds
                if "SQ_INSTS_VALU" in c:
                    roofline["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
                    # Vector issue, priced per instruction class at the rates measured in isolation on this chip
                    # (profiles/r1_valu_issue_rates.txt, two or more waves per SIMD): a two-operand instruction 2.7 cycles of its
                    # SIMD, a three-source one (the fused multiply-adds of the node scan) 4.3, a transcendental one four times a
                    # simple one.  The counters single out only the FP classes; every other vector instruction (compares, selects,
                    # min3 / med3, moves, lane reads) is priced as a two-operand one, so the figure is a LOWER bound of the
                    # issue port's occupation - and at most 1 by construction of the rates.
                    fma, trans = c.get("SQ_INSTS_VALU_FMA_F32", 0), c.get("SQ_INSTS_VALU_TRANS_F32", 0)
                    issue_cycles = 4.3 * fma + 4 * 2.7 * trans + 2.7 * (c["SQ_INSTS_VALU"] - fma - trans)
                    roofline["valu_issue_busy_min"] = issue_cycles / cycles
                    roofline["valu_issue_busy_min_is"] = "sum over instruction classes of count x cycles measured per class (profiles/r1_valu_issue_rates.txt) / (SIMDs x kernel time x 2.4 GHz); unclassified instructions priced at the cheapest class: a lower bound"
                if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES"):
                    roofline["wave_cycles_waiting_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]   # wave-cycles spent waiting for any instruction to complete or issue
                if "SQ_INSTS_SALU" in c:
                    roofline["salu_smem_insts_per_launch"] = c["SQ_INSTS_SALU"] + c.get("SQ_INSTS_SMEM", 0)
                    roofline["scalar_issue_busy"] = roofline["salu_smem_insts_per_launch"] * 4 / cycles   # one scalar issue per SIMD every 4 cycles
        if roofline and roofline.get("traffic"):
            roofline["hbm_achieved_GBps"] = roofline["traffic"] / ((ms_per_step if two_deep else kernel_ms) * 1e-3) / 1e9
            roofline["hbm_frac"] = roofline["hbm_achieved_GBps"] / PEAK_HBM_GBS
        out = {
            "metric": "Mrays/s at %dx%d %s" % (W, H, os.path.splitext(os.path.basename(args.scene))[0]), "value": mrays, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic subdivided icosahedron over the reference's plane, camera and lights (deterministic)" if base_dir else
                     "the reference's own scene and mesh files (deterministic, no RNG anywhere in the path)"),
            "config": {"workload": "%s %dx%d depth %d" % (os.path.join("scenes", os.path.basename(args.scene)), W, H, args.depth), "rays_per_frame": rays_total,
                       "shadow_rays_per_frame": shadow_total, "triangles": info["n_triangles"], "octree_nodes": info["n_nodes"],
                       "partition": ("one GPU renders the whole frame, no exchange" if not dist_on else
                                     "%d-row bands round-robin over %d GPU(s), one RCCL gather per frame%s" % (
                                         BAND_ROWS, world, ", %d frames in flight (the render of frame k+1 overlaps the end and the gather of frame k)" % in_flight
                                         if two_deep else " overlapped with the next frame's render")),
                       "library": rt.version()},
            "roofline": roofline,
        }
        if single:
            # `value` above is the throughput with frames in flight; this is one frame at a time, first launch -> last tile on rank 0
            out["value_is"] = "throughput with %d frames in flight per GPU" % in_flight if two_deep else "frames one at a time"
            out["single_frame"] = {"ms_per_frame_latency": single[0] * 1e3 / args.steps, "Mrays_per_s": rays_total / (single[0] / args.steps) / 1e6,
                                   "kernel_ms": single[1]}
        else:
            out["value_is"] = "frames back to back on one stream (one frame in flight)"
        if world == 1 and not args.no_cpu_baseline:
            threads = min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), 16)
            out["cpu_baseline"] = cpu_baseline(args, threads)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
