#!/bin/bash
# PMC passes characterising the instruction mix of the pipeline kernels (SQ counters only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_mix
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/prof_render.py 1920 1080 16 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/prof_render.py 1920 1080 16 > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/p3 -- python3 $R/tools/prof_render.py 1920 1080 16 > $O/p3.log 2>&1
find $O -name "*counter_collection.csv" | head
