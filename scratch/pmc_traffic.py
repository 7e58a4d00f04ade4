"""Sum FETCH_SIZE / WRITE_SIZE per kernel from two rocprofv3 --pmc passes of bench.py (diagnostic)."""
import csv, glob, sys, collections, json
def load(d):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            k = ("attn_spatial" if "attn_spatial_kernel" in k else "attn_combine" if "attn_combine" in k else "gemm_dma" if "gemm_dma_kernel" in k
                 else "conv_dma" if "conv3_dma" in k else "gemm" if "gemm_kernel" in k else "layernorm" if "layernorm" in k else None)
            if k is None: continue
            a = agg[(k, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
out = {}
for d in sys.argv[1:]:
    for (k, c), (n, v) in load(d).items():
        out.setdefault(k, {})[c] = {"launches": n, "mean_per_launch": v / n}
print(json.dumps(out, indent=1))
