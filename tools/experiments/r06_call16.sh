#!/bin/bash
# round-6 call 16: the evidence set r06b on the final tree — full GPU suite, smoke, bench lines, kernel traces (headline and IISPT), network counters
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=r06b
O=$R/gpurun_out/${TAG}_evidence
mkdir -p $O
cd $R
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $O/${TAG}_pytest_gpu.txt 2>&1; tail -5 $O/${TAG}_pytest_gpu.txt
( time timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/${TAG}_smoke.txt 2>&1; tail -4 $O/${TAG}_smoke.txt
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/path_stats -- python3 $R/bench.py --steps 10 --warmup 3 --sub-configs none --cpu-seconds 0 > $O/path_stats.log 2>&1 )
find $O/path_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_path_kernel_stats.csv
tail -1 $O/path_stats.log | cut -c1-300
bash tools/experiments/r06_call8.sh $TAG
