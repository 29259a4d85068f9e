#!/bin/bash
# round-5 call 2: the hand-written network against the CPU module on the timed batch (who is off, MIOpen or the HIP path?),
# then a kernel trace of the HIP network alone (per-layer times)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_call2
mkdir -p $O
cd $R
timeout 600 python tools/net_check.py 8192 > $O/net_check.txt 2>&1; tail -2 $O/net_check.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/net_check.py 8192 --no-torch > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -30 $O/kernel_stats.csv | cut -c1-220
