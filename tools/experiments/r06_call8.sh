#!/bin/bash
# round-6 call 8: the evidence set r06a — the bench line with its sub-blocks, the IISPT line, the network's kernel trace, matrix-pipe / LDS counters and traffic
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-r06a}
O=$R/gpurun_out/${TAG}_evidence
mkdir -p $O
cd $R
( time timeout 900 python3 bench.py --steps 20 --warmup 3 ) > $O/${TAG}_bench.json 2> $O/bench.err; tail -4 $O/bench.err
timeout 600 python3 bench.py --workload iispt --steps 5 --warmup 2 > $O/${TAG}_bench_iispt.json 2> $O/bench_iispt.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iispt_stats -- python3 $R/bench.py --workload iispt --steps 3 --warmup 2 --cpu-seconds 0 > $O/iispt_stats.log 2>&1 )
find $O/iispt_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_iispt_kernel_stats.csv
bash tools/net_pmc.sh default > $O/net_pmc.log 2>&1; cp gpurun_out/net_pmc/default.txt $O/${TAG}_net_pmc.txt
bash tools/net_traffic.sh > $O/net_traffic.log 2>&1; cp gpurun_out/net_traffic/traffic.json $O/${TAG}_net_traffic.json
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python3 bench.py --workload iispt --steps 3 --warmup 2 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(json.dumps({'run': $i, 'ms_per_step': j['ms_per_step'], 'network_ms': j['stage_ms_per_step']['network'], 'probes_per_s': j['value']}))"; done > $O/${TAG}_iispt_ten_processes.jsonl
python3 - <<PY
import json
j = json.loads(open('$O/${TAG}_bench.json').readline())
print({k: j[k] for k in ('value', 'ms_per_step')}, j['roofline']['frac'], j['cpu_baseline']['value'])
for k, b in j['configs'].items():
    print(k, b['ms_per_step'], b['value'], b['wall_seconds_of_this_block'], b['roofline'].get('frac'), b['roofline'].get('frac_executed'))
i = json.loads(open('$O/${TAG}_bench_iispt.json').readline())
print(i['ms_per_step'], i['stage_ms_per_step'], i['roofline']['frac_executed'], i['roofline']['agreement_with_the_module'])
PY
cat $O/${TAG}_net_pmc.txt; tail -1 $O/net_traffic.log; cat $O/${TAG}_iispt_ten_processes.jsonl | head -3
