#!/usr/bin/env python3
"""Turns the rocprofv3 passes of tools/profile_headline.sh TAG (gpurun_out/prof_TAG, gpurun_out/summary_TAG.txt) into the files
kept under profiles/: rROUND_TAG_pmc.txt, rROUND_TAG_kernel_stats.txt and rROUND_TAG_counters.json.  The JSON carries the digest
of the kernel sources (raytracer_amd.kernel_source_sha256): bench.py quotes its numbers only while that digest matches.
   python3 tools/profile_to_json.py TAG [ROUND]"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt   # noqa: E402


def main():
    tag = sys.argv[1]
    rnd = sys.argv[2] if len(sys.argv) > 2 else "2"
    prof = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    summary = open(os.path.join(ROOT, "gpurun_out", "summary_%s.txt" % tag)).read()
    counters = {}
    kernel = None
    for line in summary.splitlines():
        m = re.match(r"\s+(\w+)\s+(\d+) per dispatch", line)
        if m:
            counters[m.group(1)] = int(m.group(2))
        elif "dispatches=" in line and kernel is None:
            kernel = line.split("  dispatches")[0].strip()
    stats = glob.glob(os.path.join(prof, "trace", "*", "*kernel_stats.csv"))
    avg_ns = calls = None
    if stats:
        for r in csv.DictReader(open(stats[0])):
            if "render_wave" in r["Name"]:
                avg_ns, calls = float(r["AverageNs"]), int(r["Calls"])
                m = re.search(r"render_wave<[^>]*>", r["Name"])
                kernel = m.group(0) if m else r["Name"]
                break
    base = os.path.join(ROOT, "profiles", "r%s_%s" % (rnd, tag))
    with open(base + "_pmc.txt", "w") as f:
        f.write("# rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline   (%s; tools/profile_headline.sh)\n" % rt.version() if False else
                "# rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline   (tools/profile_headline.sh %s)\n" % tag)
        f.write("# one pass per block below; values are per dispatch of the render kernel (one 3840x2160 meshDragon frame); FETCH_SIZE / WRITE_SIZE in KB as reported;\n")
        f.write("# the vgpr field of rocprofv3 reads half the per-lane register count of a wave64 kernel (40 for the 80-register variant)\n")
        f.write(summary.split("## kernel trace")[0])
    if stats:
        with open(base + "_kernel_stats.txt", "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline\n")
            f.write(open(stats[0]).read())
    lanes = 64
    fp = {k: counters.get("SQ_INSTS_VALU_" + k, 0) for k in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")}
    executed_flop = lanes * (fp["ADD_F32"] + fp["MUL_F32"] + 2 * fp["FMA_F32"] + fp["TRANS_F32"])
    out = {
        "source_sha256": rt.kernel_source_sha256(),
        "workload": "scenes/meshDragon.json 3840x2160 depth 3",
        "kernel": kernel, "kernel_avg_ns_under_trace": avg_ns, "kernel_calls_under_trace": calls,
        "how": "tools/profile_headline.sh %s: one rocprofv3 --pmc pass per counter group over bench.py --steps 5 --warmup 2, kernel trace over --steps 20 --warmup 3; "
               "per dispatch of the render kernel" % tag,
        "counters_per_launch": counters,
        "executed_fp32_flop_per_launch": executed_flop,
        "fetch_size_kb": counters.get("FETCH_SIZE"), "write_size_kb": counters.get("WRITE_SIZE"),
        "traffic_bytes_per_launch": (counters.get("FETCH_SIZE", 0) + counters.get("WRITE_SIZE", 0)) * 1024 if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters else None,
        "note": "FETCH_SIZE is reported raw (the gfx950 x2 correction is calibrated for wide coalesced streaming reads; this kernel reads records through the scalar "
                "cache and 16-byte gathers): with the correction applied traffic would be FETCH_SIZE KB more",
    }
    with open(base + "_counters.json", "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", base + "_{pmc.txt,kernel_stats.txt,counters.json}")


if __name__ == "__main__":
    main()
