#!/usr/bin/env python3
"""Turns the rocprofv3 passes of tools/profile_scene_all.sh TAG SCENE (gpurun_out/prof_TAG, summary_TAG.txt, bench_TAG.json) into
the files kept under profiles/: rROUND_TAG_pmc.txt, rROUND_TAG_kernel_stats.txt, rROUND_TAG_counters.json, rROUND_TAG_bench.json.
The JSON carries the digest of the kernel sources and the workload; bench.py --scene quotes its numbers only while both match.
Counters are per FRAME: summed over the dispatches of every kernel of the frame (the instrumented variant's one launch, which
bench.py makes outside the timed region to get the test counts, is left out) and divided by the frames the profiled command rendered.
   python3 tools/profile_scene_to_json.py TAG ROUND FRAMES     (FRAMES = 1 + warmup + steps of the profiled bench.py run: 8)"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt   # noqa: E402


def is_instrumented(name):
    m = re.search(r"render_wave<(\w+), (\w+)", name)
    return bool(m and m.group(2) == "true") or "render_tiles" in name


def main():
    tag, rnd, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    prof = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    bench = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_%s.json" % tag)))
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    dispatches = collections.defaultdict(lambda: collections.defaultdict(set))
    saw_instrumented = False
    for pass_dir in sorted(glob.glob(os.path.join(prof, "pmc_*"))):
        if not os.path.isdir(pass_dir):
            continue
        found = sorted(glob.glob(os.path.join(pass_dir, "*", "*counter_collection.csv")), key=os.path.getmtime)
        if not found:
            continue
        for r in csv.DictReader(open(found[-1])):   # (the newest run of the pass, should an older one still lie there)
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if is_instrumented(k) or not re.match(r"render_wave|q_|unpack_bands", k):
                saw_instrumented = saw_instrumented or is_instrumented(k)
                continue
            per_kernel[k][r["Counter_Name"]] += float(r["Counter_Value"])
            dispatches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    frame = collections.defaultdict(float)
    kernels = {}
    # (the queue-ordering and summing kernels have no instrumented variant of their own: the one instrumented frame that
    #  bench.py renders outside its timed region launches them too, so they ran for one frame more than the render kernels)
    counted_frame = 1 if saw_instrumented else 0
    for k, c in per_kernel.items():
        n = max(len(v) for v in dispatches[k].values())
        f_k = frames + (counted_frame if k.startswith("q_") else 0)
        for name in list(c):
            c[name] *= frames / f_k
        kernels[k] = {"dispatches_per_frame": n / f_k, "counters_per_dispatch": {name: int(v * f_k / frames / len(dispatches[k][name])) for name, v in sorted(c.items())}}
        for name, v in c.items():
            frame[name] += v / frames
    stats = sorted(glob.glob(os.path.join(prof, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1:]
    trace = []
    if stats:
        for r in csv.DictReader(open(stats[0])):
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            trace.append({"kernel": name, "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"])})
    base = os.path.join(ROOT, "profiles", "r%s_%s" % (rnd, tag))
    summary = open(os.path.join(ROOT, "gpurun_out", "summary_%s.txt" % tag)).read()
    with open(base + "_pmc.txt", "w") as f:
        f.write("# rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --scene %s --steps 5 --warmup 2 --no-cpu-baseline   (tools/profile_scene_all.sh %s)\n"
                % (bench["config"]["workload"].split()[0], tag))
        f.write("# one pass per block below; values are per dispatch of each kernel; FETCH_SIZE / WRITE_SIZE in KB as reported\n")
        f.write(summary.split("## kernel trace")[0])
    if stats:
        with open(base + "_kernel_stats.txt", "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --scene ... --steps 20 --warmup 3 --no-cpu-baseline\n")
            f.write(open(stats[0]).read())
    fp = {k: frame.get("SQ_INSTS_VALU_" + k, 0) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")}
    out = {
        "source_sha256": rt.kernel_source_sha256(),
        "workload": bench["config"]["workload"],
        "frames_profiled": frames,
        "how": "tools/profile_scene_all.sh %s: one rocprofv3 --pmc pass per counter group over bench.py --scene --steps 5 --warmup 2, kernel trace over --steps 20 --warmup 3" % tag,
        "counters_per_frame": {k: int(v) for k, v in sorted(frame.items())},
        "counters_per_launch": {k: int(v) for k, v in sorted(frame.items())},   # (the key bench.py reads; for a frame of several launches: per frame)
        "kernels": kernels,
        "kernel_trace": trace,
        "executed_fp32_flop_per_launch": int(64 * (fp["ADD_F32"] + fp["MUL_F32"] + 2 * fp["FMA_F32"] + fp["TRANS_F32"])),
        "fetch_size_kb": int(frame.get("FETCH_SIZE", 0)), "write_size_kb": int(frame.get("WRITE_SIZE", 0)),
        "traffic_bytes_per_launch": int((frame.get("FETCH_SIZE", 0) + frame.get("WRITE_SIZE", 0)) * 1024),
        "bench_line_of_the_same_run": {"ms_per_step": bench["ms_per_step"], "value": bench["value"], "roofline_frac": (bench.get("roofline") or {}).get("frac")},
        "note": "FETCH_SIZE is reported raw (without the gfx950 x2 correction, which is calibrated for wide coalesced streaming reads)",
    }
    with open(base + "_counters.json", "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    with open(base + "_bench.json", "w") as f:
        json.dump(bench, f)
        f.write("\n")
    print("wrote", base + "_{pmc.txt,kernel_stats.txt,counters.json,bench.json}")


if __name__ == "__main__":
    main()
