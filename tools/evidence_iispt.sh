#!/bin/bash
# BASELINE config 5's evidence alone (tools/evidence.sh collects it with everything else): tools/evidence_iispt.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r05d}
O=$R/gpurun_out/${TAG}_evidence_iispt
mkdir -p $O
cd $R
timeout 600 python3 bench.py --workload iispt --steps 5 --warmup 2 > $O/${TAG}_bench_iispt.json 2> $O/bench_iispt.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iispt_stats -- python3 $R/bench.py --workload iispt --steps 3 --warmup 2 --cpu-seconds 0 > $O/iispt_stats.log 2>&1 )
find $O/iispt_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_iispt_kernel_stats.csv
bash tools/net_pmc.sh default > $O/net_pmc.log 2>&1; cp gpurun_out/net_pmc/default.txt $O/${TAG}_net_pmc.txt
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python3 bench.py --workload iispt --steps 3 --warmup 2 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(json.dumps({'run': $i, 'ms_per_step': j['ms_per_step'], 'network_ms': j['stage_ms_per_step']['network'], 'probes_per_s': j['value']}))"; done > $O/${TAG}_iispt_ten_processes.jsonl
timeout 600 python -m pytest tests/test_iispt_gather.py tests/test_iispt_direct.py tests/test_iispt_nn.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
head -c 400 $O/${TAG}_bench_iispt.json; echo; cat $O/${TAG}_iispt_ten_processes.jsonl
