#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_room
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --workload boxroom --steps 1 --warmup 0 --cpu-seconds 0 --other-steps 0 --alone-steps 0 > $O/sq.log 2>&1
timeout 600 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -- python3 $R/bench.py --workload boxroom --steps 1 --warmup 0 --cpu-seconds 0 --other-steps 0 --alone-steps 0 > $O/tcc.log 2>&1
python3 - <<PY
import csv,glob,collections
for sub in ('sq','tcc'):
    f=glob.glob('$O/%s/*/*counter_collection.csv'%sub)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0].replace('void iile::','')
        if '<true' in n: continue
        agg[n.split('<')[0]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if k in ('k_extend','k_shade','k_shadow'):
            if sub=='sq':
                lu=v['SQ_THREAD_CYCLES_VALU']/(64*v['SQ_ACTIVE_INST_VALU']); vb=4*v['SQ_ACTIVE_INST_VALU']/1024/(v['SQ_BUSY_CYCLES']/32)
                print(k,'VALU %.2fe9'%(v['SQ_INSTS_VALU']/1e9),'lane_util %.3f'%lu,'valu_busy %.3f'%vb)
            else:
                print(k,'L2 req %.2fe9 hit rate %.3f'%(v['TCC_REQ_sum']/1e9, v['TCC_HIT_sum']/max(v['TCC_REQ_sum'],1)))
PY
