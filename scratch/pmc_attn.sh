#!/bin/bash
# MFMA-pipe and wait counters of the attention kernel, round-1 form (EDV_ATTN_PIPE=0) vs the pipelined one; run on the GPU box from the repo root
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-pmc_attn}
mkdir -p $O
for v in 0 1; do
  (cd /tmp && EDV_ATTN_PIPE=$v timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $O/p$v -o k -- python3 $R/scratch/attn_prof.py ${2:-8} 1370 6 > $O/p$v.log 2>&1)
  python3 - $O/p$v $v <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
rows = []
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True): rows += list(csv.DictReader(open(fn)))
trace = []
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True): trace += list(csv.DictReader(open(fn)))
dur = collections.defaultdict(list)
for r in trace: dur[r["Kernel_Name"][:50]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "attn" not in k: continue
    m = {n: sum(x) / len(x) for n, x in c.items()}
    us = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
    print(f"pipe={v} {k}: {us:.1f} us; MFMA busy / SQ busy x32 = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['SQ_BUSY_CYCLES'] * 32):.3f}; clock (GUI_ACTIVE/8/dur) = {m['GRBM_GUI_ACTIVE'] / 8 / (us * 1e3):.3f} GHz; "
          f"per wave-cycle: wait_any {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f} wait_inst {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f} active {m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f}; LDS conflicts {m['SQ_LDS_BANK_CONFLICT']:.0f}")
PY
done
