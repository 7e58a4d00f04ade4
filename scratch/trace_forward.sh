#!/bin/bash
# Per-launch timeline of ONE forward (name, duration, gap before it): rocprofv3 --kernel-trace of a short single-stream run, last forward listed in order.
# usage: scratch/trace_forward.sh <outdir under gpurun_out> [bench args]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; shift
mkdir -p $O
(cd /tmp && EDV_HEAD_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o k -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --in-flight 1 --steps 3 --warmup 1 "$@" > $O/trace.log 2>&1)
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/forward_timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# forwards start at a patchify_kernel: take the 4th (create, warm-up, then the timed steps) up to the next one
starts = [i for i, n in enumerate(names) if "patchify_kernel" in n]
start, end = starts[3], starts[4]
prev_end = None
tot = gap_tot = 0
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:60]
for i in range(start, end):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{i - start:4d} {short(names[i]):60s} {(e - s) / 1e3:9.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', '?'):>8s} wg {r.get('Workgroup_Size_X', '?')}")
    tot += (e - s) / 1e3
    gap_tot += max(gap, 0)
    prev_end = e
print(f"launches {end - start}, kernel time {tot:.1f} us, gaps {gap_tot:.1f} us, span {(int(rows[end - 1]['End_Timestamp']) - int(rows[start]['Start_Timestamp'])) / 1e3:.1f} us")
PY
rm -rf $O/trace
tail -1 $O/forward_timeline.txt
