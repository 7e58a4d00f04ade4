#!/bin/bash
# HBM-side traffic of the dominant kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the headline bench command,
# summed per kernel class and written as profiles/r02_{gemm,attn,hbm}_traffic.json.  Run on the GPU box from the repo root.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-pmc_traffic}
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o k -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/$c.log 2>&1) || echo "pmc $c failed"
done
python3 - $O $R <<'PY'
import csv, glob, sys, collections, json, os
O, R = sys.argv[1], sys.argv[2]
def cls(k):
    if "gemm_dma_kernel" in k: return "gemm"
    if "attn_lean_kernel" in k or "attn_spatial_kernel" in k: return "attn"
    if "attn_combine" in k: return "attn_combine"
    if "layernorm_kernel" in k: return "layernorm"
    if "groupnorm" in k: return "groupnorm"
    if "bilinear" in k: return "bilinear"
    if "geglu" in k: return "geglu"
    if "dot_channels" in k: return "dot_channels"
    if "patchify" in k: return "patchify"
    if "conv3_dma" in k: return "conv3x3"
    return None
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = cls(r["Kernel_Name"])
            if k and r["Counter_Name"] == c:
                a = agg[k][c]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for fn in glob.glob(f"{O}/{c}/**/*counter_collection.csv", recursive=True):
        os.system(f"cp {fn} {R}/profiles/r02_pmc_{c.lower()}_bench_T8.csv")
def per_launch(k):
    f, w = agg[k]["FETCH_SIZE"], agg[k]["WRITE_SIZE"]
    if not f[0] or not w[0]: return None
    return {"launches": f[0], "FETCH_SIZE_kb_per_launch": round(f[1] / f[0], 1), "WRITE_SIZE_kb_per_launch": round(w[1] / w[0], 1),
            "traffic_bytes_per_launch": int((2 * f[1] / f[0] + w[1] / w[0]) * 1024)}
cfg = {"encoder": "vits", "T": 8, "image": 518}
cmd = "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events"
corr = "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact; counter unit KB"
raw = ["profiles/r02_pmc_fetch_size_bench_T8.csv", "profiles/r02_pmc_write_size_bench_T8.csv"]
g = per_launch("gemm")
if g:
    json.dump({"kernel": "gemm_dma_kernel (every dense F.linear / 1x1-conv launch of one forward, all shapes)", "command": cmd, "raw": raw, "config": cfg, **g,
               "correction": corr}, open(f"{R}/profiles/r02_gemm_traffic.json", "w"), indent=1)
a, ac = per_launch("attn"), per_launch("attn_combine")
if a:
    tot = a["traffic_bytes_per_launch"] + (ac["traffic_bytes_per_launch"] if ac else 0)
    json.dump({"kernel": "attn_lean_kernel + attn_combine_kernel (one encoder-block attention call)", "command": cmd, "raw": raw, "config": cfg, "attn": a, "combine": ac,
               "traffic_bytes_per_launch": tot, "correction": corr}, open(f"{R}/profiles/r02_attn_traffic.json", "w"), indent=1)
hb = {k: per_launch(k) for k in ("layernorm", "groupnorm", "bilinear", "geglu", "dot_channels", "patchify")}
# groupnorm: bench brackets one call = its statistics + apply launches; sum the call's launches
json.dump({"kernels": hb, "traffic_bytes_per_launch": {k: (v["traffic_bytes_per_launch"] if v else None) for k, v in hb.items()}, "command": cmd, "raw": raw, "config": cfg,
           "correction": corr, "note": "per kernel LAUNCH (groupnorm: mean over its 2-3 launches per call; bilinear: mean over every resample of the head)"},
          open(f"{R}/profiles/r02_hbm_traffic.json", "w"), indent=1)
print(json.dumps({"gemm": g, "attn": a, "combine": ac, "hbm": {k: (v["traffic_bytes_per_launch"] if v else None) for k, v in hb.items()}}, indent=1))
PY
cp $R/profiles/r02_*_traffic.json $R/profiles/r02_pmc_*_bench_T8.csv $O/ 2>/dev/null
