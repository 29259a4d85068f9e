#!/bin/bash
# round-6 call 5: is the pipeline's ray-queue order worth sorting? (tools/coherence_probe.py pixels: the room's second-bounce rays in
# path order, sorted by (cell, octant), sorted by the full origin code, shuffled — k_trace durations from the kernel trace)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for scene in boxroom; do
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$scene -- python3 $R/tools/coherence_probe.py 8000000 $scene pixels > $O/probe_$scene.txt 2> $O/probe_$scene.err
  cat $O/probe_$scene.txt
  python3 - <<PY
import csv, glob
f = glob.glob('$O/trace_$scene/**/*kernel_trace.csv', recursive=True)
rows = [r for f_ in f for r in csv.DictReader(open(f_)) if 'k_trace' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print("$scene k_trace launches (ms; first wave, then per bounce: warm-up + 3 x 6 orders):", [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 3) for r in rows])
PY
done
