import csv, glob, sys, collections
rows = collections.defaultdict(dict)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "render_wave" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[int(r["Dispatch_Id"])]["k"] = r["Kernel_Name"].split("render_wave")[1][:40]
for d in sorted(rows):
    c = rows[d]
    print(d, c["k"], " ".join("%s=%.1fM" % (k, v / 1e6) for k, v in sorted(c.items()) if k != "k"))
