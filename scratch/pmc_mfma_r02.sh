#!/bin/bash
# MFMA-pipe utilisation of the dominant kernels of the headline bench (one --pmc pass, kernel trace only) -> profiles/r02_mfma_busy.txt,
# then the attention launch alone, round-1 kernel (EDV_ATTN_LEAN=0) vs attn_lean_kernel -> profiles/r02_attn_pmc.txt.  Run on the GPU box from the repo root.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-pmc_mfma}
mkdir -p $O
(cd /tmp && EDV_HEAD_STREAMS=1 timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/bench -o k -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/bench.log 2>&1) || echo "pmc bench failed"
python3 - $O/bench > $R/profiles/r02_mfma_busy.txt <<'PY'
import csv, glob, sys, collections, re
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        m = re.search(r"(gemm_dma_kernel<[^>]*>|conv3_dma_kernel<[^>]*>|attn_lean_kernel|attn_spatial_kernel<[^>]*>)", k)
        if not m: continue
        key = (m.group(1), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_BUSY_CYCLES": agg[key]["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("# MFMA pipe utilisation of the dominant kernels, round 2 (buffer-DMA GEMM / convolution, lean attention).")
print("# EDV_HEAD_STREAMS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events")
print("# (ViT-S 518x518 T=8, one stream).  busy vs wall = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the launch's own clock count (8 XCDs);")
print("# busy vs SQ busy = the same counter against SQ_BUSY_CYCLES x 32 (summed over 32 shader engines of 32 SIMDs).  Rows: kernel instantiation x grid, >= 6 launches, by total time.")
print(f"# {'kernel':44s} {'workgroups':>10s} {'launches':>8s} {'busy vs wall':>13s} {'busy vs SQ busy':>16s} {'mean us under PMC':>18s} {'clock GHz':>10s}")
rows = []
for (k, g), c in agg.items():
    n = len(c["SQ_BUSY_CYCLES"])
    if n < 6: continue
    mf, sq, gui, du = (sum(c[x]) / n for x in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "dur"))
    rows.append((n * du, f"{k:46s} {g:10d} {n:8d} {100 * mf / (gui / 8 * 1024):12.1f}% {100 * mf / (sq * 32):15.1f}% {du / 1e3:18.1f} {gui / 8 / du:10.3f}"))
for _, line in sorted(rows, reverse=True)[:16]: print(line)
PY
cat $R/profiles/r02_mfma_busy.txt
{
echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -- python3 scratch/attn_prof.py 8 1370 6   (scratch/pmc_mfma_r02.sh)"
echo "# the T=8 ViT-S attention launch alone, 50 times; lean=0: the round-1 kernel attn_spatial_kernel<4,64> (EDV_ATTN_LEAN=0), lean=1: attn_lean_kernel (product)"
} > $R/profiles/r02_attn_pmc.txt
for v in 0 1; do
  (cd /tmp && EDV_ATTN_LEAN=$v timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $O/p$v -o k -- python3 $R/scratch/attn_prof.py 8 1370 6 > $O/p$v.log 2>&1) || echo "pmc attn $v failed"
  python3 - $O/p$v $v >> $R/profiles/r02_attn_pmc.txt <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "attn_" not in k: continue
        k = "attn_combine_kernel" if "combine" in k else ("attn_lean_kernel" if "attn_lean" in k else "attn_spatial_kernel<4,64>")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_BUSY_CYCLES": agg[k]["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in agg.items():
    if "combine" in k: continue
    m = {n: sum(x) / len(x) for n, x in c.items()}
    print(f"lean={v} {k}: {m['dur'] / 1e3:.1f} us under counters; MFMA busy / (SQ busy x 32) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['SQ_BUSY_CYCLES'] * 32):.3f}; "
          f"MFMA busy / (GUI_ACTIVE / 8 x 1024) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}; clock (GUI_ACTIVE / 8 / duration) = {m['GRBM_GUI_ACTIVE'] / 8 / m['dur']:.3f} GHz; "
          f"per wave-cycle: wait_any {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f} wait_inst {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} active {m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}; LDS bank conflicts {m['SQ_LDS_BANK_CONFLICT']:.0f}")
PY
done
cat $R/profiles/r02_attn_pmc.txt
cp $R/profiles/r02_mfma_busy.txt $R/profiles/r02_attn_pmc.txt $O/
