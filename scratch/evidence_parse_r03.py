"""Turn the rocprofv3 --pmc passes of scratch/evidence_r03.sh into <name>_traffic.json (HBM-side bytes per launch per kernel class; FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE exact, counter unit KB) and <name>_mfma_busy.txt (SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES x 32
and against the launch's wall time from GRBM_GUI_ACTIVE / 8 XCDs at the clock the launch held)."""
import collections, csv, glob, json, sys

O, name, args = sys.argv[1], sys.argv[2], sys.argv[3:]


def cls(k):
    if "gemm_x6_kernel" in k: return "gemm_x6"
    if "gemm_dma_kernel" in k: return "gemm"
    if "attn_lean_kernel" in k or "attn_spatial_kernel" in k or "attn_x6_kernel" in k: return "attn"
    if "attn_combine" in k: return "attn_combine"
    if "layernorm_kernel" in k: return "layernorm"
    if "groupnorm" in k: return "groupnorm"
    if "bilinear" in k: return "bilinear"
    if "geglu_kernel" in k: return "geglu"
    if "dot_channels" in k: return "dot_channels"
    if "patchify" in k: return "patchify"
    if "conv3_dma" in k: return "conv3x3"
    if "attn_temporal" in k: return "attn_temporal"
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(f"{O}/pmc_{name}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = cls(r["Kernel_Name"])
            if k and r["Counter_Name"] == c:
                a = agg[k][c]
                a[0] += 1
                a[1] += float(r["Counter_Value"])


def per_launch(k):
    f, w = agg[k]["FETCH_SIZE"], agg[k]["WRITE_SIZE"]
    if not f[0] or not w[0]:
        return None
    return {"launches": f[0], "FETCH_SIZE_kb_per_launch": round(f[1] / f[0], 1), "WRITE_SIZE_kb_per_launch": round(w[1] / w[0], 1),
            "traffic_bytes_per_launch": int((2 * f[1] / f[0] + w[1] / w[0]) * 1024)}


def arg(flag, default):
    return args[args.index(flag) + 1] if flag in args else default


cfg = {"encoder": arg("--encoder", "vits"), "T": int(arg("--T", "8")), "image": int(str(arg("--image", "518")).split("x")[0]), "products": arg("--products", "bf16x6")}
out = {"config": cfg, "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two separate passes) --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline "
       "--no-kernel-events --in-flight 1 --steps 4 --warmup 2 (after: " + " ".join(args) + ")",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact; counter unit KB",
       "per_class": {k: per_launch(k) for k in ("gemm", "gemm_x6", "attn", "attn_combine", "conv3x3", "layernorm", "groupnorm", "bilinear", "geglu", "dot_channels", "patchify", "attn_temporal")}}
g, a, ac = out["per_class"]["gemm"], out["per_class"]["attn"], out["per_class"]["attn_combine"]
out["gemm_traffic_bytes_per_launch"] = g["traffic_bytes_per_launch"] if g else None
out["attn_call_traffic_bytes"] = (a["traffic_bytes_per_launch"] + (ac["traffic_bytes_per_launch"] if ac else 0)) if a else None
json.dump(out, open(f"{O}/{name}_traffic.json", "w"), indent=1)

# ---- MFMA-pipe utilisation ----
def short(kn):
    kn = kn.replace("void ", "").replace("edv::(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(kn):  # cut the argument list: the first "(" outside the template brackets
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0:
            return kn[:i]
    return kn


rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for fn in glob.glob(f"{O}/pmc_{name}_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        key = short(r["Kernel_Name"])
        rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            calls[key] += 1
dur = collections.defaultdict(float)
for fn in glob.glob(f"{O}/pmc_{name}_mfma/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
with open(f"{O}/{name}_mfma_busy.txt", "w") as f:
    f.write(f"MFMA-pipe utilisation by counters, {name}: one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) of\n"
            f"python3 bench.py {' '.join(args)} --no-cpu-baseline --no-kernel-events --in-flight 1 --steps 4 --warmup 2\n"
            "busy/SQ   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32)        (share of the time a SIMD had waves in which its matrix pipe was executing)\n"
            "busy/wall = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)  (share of the launch's whole duration, ramp and tail included)\n"
            "clock     = GRBM_GUI_ACTIVE / 8 / kernel duration (reads high on launches under ~0.3 ms: MI355X_MICROARCH.md, DVFS give-back)\n\n")
    f.write(f"{'kernel':58s} {'calls':>6s} {'avg us':>9s} {'busy/SQ':>8s} {'busy/wall':>9s} {'clock GHz':>9s}\n")
    order = sorted(rows, key=lambda k: -dur[k])
    for k in order[:18]:
        v = rows[k]
        if not v["SQ_BUSY_CYCLES"] or not v["GRBM_GUI_ACTIVE"] or not v["SQ_VALU_MFMA_BUSY_CYCLES"]:
            continue
        f.write(f"{k[:58]:58s} {calls[k]:6d} {dur[k] / max(calls[k], 1) * 1e6:9.1f} {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['SQ_BUSY_CYCLES'] * 32):8.3f} "
                f"{v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['GRBM_GUI_ACTIVE'] / 8):9.3f} {v['GRBM_GUI_ACTIVE'] / 8 / dur[k] / 1e9:9.2f}\n")
print(open(f"{O}/{name}_mfma_busy.txt").read())
print(json.dumps({k: (v["traffic_bytes_per_launch"] if v else None) for k, v in out["per_class"].items()}))
