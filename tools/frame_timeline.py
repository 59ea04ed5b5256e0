#!/usr/bin/env python3
"""The launches of the last frame of a rocprofv3 kernel trace, in order, with their durations and the gaps between them:
   python3 tools/frame_timeline.py KERNEL_TRACE.csv
A frame rendered from queues is taken from its primary-ray pass to the kernel that adds up the term slots."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    resolves = [i for i, n in enumerate(names) if "q_resolve" in n]
    waves = [i for i, n in enumerate(names) if "render_wave" in n]
    if not waves:
        raise SystemExit("no render_wave launch in the trace")
    if resolves:   # a frame rendered from queues ends with the kernel that adds up the term slots
        end = resolves[-1]
        start = max(i for i in waves if i < end and (i == 0 or "q_scatter" not in names[i - 1]))   # its primary-ray pass
    else:
        start = end = waves[-1]
    t0 = int(rows[start]["Start_Timestamp"])
    prev = None
    print("%-44s %10s %10s %8s %12s" % ("kernel", "start_us", "dur_us", "gap_us", "grid"))
    for r in rows[start:end + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
        print("%-44s %10.1f %10.1f %8.1f %12s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get("Grid_Size_X", "")))
        prev = e
    print("frame: %.1f us from the first launch to the end of the last" % ((prev - t0) / 1e3))


if __name__ == "__main__":
    main()
