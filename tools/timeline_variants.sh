#!/bin/bash
# first-bounce kernel durations for each named variant (rocprofv3 --kernel-trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  O=$R/gpurun_out/tlv_$n; rm -rf $O; mkdir -p $O
  if [ "$n" = base ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$n.so; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/prof_render.py 1920 1080 64 2 > $O/t.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('$O/t/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last plain pass starts at the last bounce-0 k_extend of the product build (camera rays are made inside it: template
# arguments <false, ., true>); passes that launch k_generate (instrumented / explicit lists) are not the timed ones
gen=[i for i,r in enumerate(rows) if 'k_extend<false' in r['Kernel_Name'] and r['Kernel_Name'].split('(')[0].rstrip().endswith('true>')]
idx=gen[-1] if gen else 0
out=[]
for r in rows[idx:idx+12]:
    n=r['Kernel_Name'].split('(')[0].replace('void iile::','').replace('iile::','')
    out.append('%s %.2f'%(n.replace('<false>',''),(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
print('$n', ' | '.join(out))
PY
done
