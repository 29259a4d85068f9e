#!/bin/bash
# start / end of every kernel of the last plain pass of one bench step, relative to the pass's first kernel
# (rocprofv3 --kernel-trace): shows what overlaps with what in the two-stream schedule and where the GPU waits
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/timeline
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --steps 2 --warmup 0 --cpu-seconds 0 --other-steps 0 --alone-steps 0 > $O/t.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$O/t/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
gen=[i for i,r in enumerate(rows) if 'k_extend<false' in r['Kernel_Name'] and r['Kernel_Name'].split('(')[0].rstrip().endswith('true>')]
idx=gen[-1] if gen else 0
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    n=r['Kernel_Name'].split('(')[0].replace('void iile::','').replace('iile::','')
    s=(int(r['Start_Timestamp'])-t0)/1e6; e=(int(r['End_Timestamp'])-t0)/1e6
    print('%-28s q%-3s %8.3f -> %8.3f  (%.3f)'%(n[:28], r.get('Queue_Id','?'), s, e, e-s))
PY
