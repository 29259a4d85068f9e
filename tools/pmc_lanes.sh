#!/bin/bash
# VALU lane utilisation / VALU busy per dispatch (two SQ counter passes); prints a table.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_lanes
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/prof_render.py 1920 1080 16 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p3 -- python3 $R/tools/prof_render.py 1920 1080 16 > $O/p3.log 2>&1
python3 - <<PY
import csv,glob,collections
def load(p):
    f=glob.glob('$O/'+p+'/*/*counter_collection.csv')[0]
    d=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k=(int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0].replace('void iile::','').replace('iile::',''))
        d.setdefault(k,{})[r['Counter_Name']]=float(r['Counter_Value'])
    return d
p1=load('p1'); p3=load('p3')
for a,b in zip(sorted(p1),sorted(p3)):
    v=p1[a]; w=p3[b]
    if a[1]!=b[1] or v['SQ_INSTS_VALU']<1e6: continue
    print(a[0], a[1][:22], 'valu=%.3e'%v['SQ_INSTS_VALU'], 'lane_util=%.2f'%(w['SQ_THREAD_CYCLES_VALU']/(v['SQ_ACTIVE_INST_VALU']*64)), 'busy_cyc=%.2e'%v['SQ_BUSY_CYCLES'], 'valu_busy=%.2f'%(v['SQ_ACTIVE_INST_VALU']*4/1024/(v['SQ_BUSY_CYCLES']/32)))
PY
