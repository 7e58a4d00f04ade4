import csv, sys, collections, glob
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
rows = []
for fn in f: rows += list(csv.DictReader(open(fn)))
agg = collections.OrderedDict()
for r in rows:
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:60], r["Grid_Size"])
    agg.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, c in agg.items():
    if "gemm" not in k[1] and "attn" not in k[1]: continue
    print(k, " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
