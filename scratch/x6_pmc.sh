#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp
O=$R/gpurun_out/x6pmc; mkdir -p $O
cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p -o k -- python3 $R/scratch/x6_pmc.py > $O/log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/p/**/*counter_collection.csv", recursive=True)[0]
kt=glob.glob("$O/p/**/*kernel_trace.csv", recursive=True)[0]
dur={r["Dispatch_Id"]:(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
rows=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k=(r["Dispatch_Id"], r["Kernel_Name"][:60])
    rows.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
seen=collections.Counter()
for (d,k),c in rows.items():
    if "gemm" not in k: continue
    seen[k]+=1
    if seen[k] not in (3,9,15): continue
    w=c["SQ_WAVE_CYCLES"]
    print(f"{k[:58]:58s} {dur[d]/1e3:8.1f}us clk={c['GRBM_GUI_ACTIVE']/8/dur[d]:.2f}GHz mfma/sqbusy={c['SQ_VALU_MFMA_BUSY_CYCLES']/c['SQ_BUSY_CYCLES']/4:.3f} wait_any={c['SQ_WAIT_ANY']/w:.2f} wait_inst={c['SQ_WAIT_INST_ANY']/w:.2f} active={c['SQ_ACTIVE_INST_ANY']/w:.2f} wait_lds={c['SQ_WAIT_INST_LDS']/w:.3f} bankconf={c['SQ_LDS_BANK_CONFLICT']/c['SQ_BUSY_CYCLES']:.3f}")
PY
rm -rf $O/p
