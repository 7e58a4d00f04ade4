import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]:
            print(f.split("/")[-2] if "/" in f else f, r["Name"].split("::")[-1][:40], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "min %.1f" % (float(r["MinNs"]) / 1e3))
