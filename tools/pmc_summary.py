#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters (CSV output) per kernel over its dispatches: `pmc_summary.py DIR [kernel-regex]`."""
import collections
import csv
import glob
import re
import sys


def main(d, needle="render_"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    for path in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if re.search(needle, r["Kernel_Name"]):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                ndisp[k].add(r["Dispatch_Id"])
                agg[k]["_vgpr"] = float(r["VGPR_Count"]); agg[k]["_sgpr"] = float(r["SGPR_Count"])
    for k, c in agg.items():
        n = len(ndisp[k])
        print("%s  dispatches=%d vgpr=%d sgpr=%d" % (k, n, c.pop("_vgpr"), c.pop("_sgpr")))
        for name in sorted(c):
            print("   %-28s %18.0f per dispatch" % (name, c[name] / n))


if __name__ == "__main__":
    main(*sys.argv[1:])
