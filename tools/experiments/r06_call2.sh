#!/bin/bash
# round-6 call 2: the split-fp16 network — its parity tests at the per-element tolerance, the frame at several network batch sizes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call2
mkdir -p $O
cd $R
( time timeout 1200 python3 -m pytest tests/test_iispt_nn.py tests/test_iispt_gather.py tests/test_iispt_host.py tests/test_abi.py -m gpu -x -q -s ) > $O/tests.txt 2>&1; grep -n "passed\|failed\|within 1e-4\|max error" $O/tests.txt | head -30
for nb in 8192 16384 32768; do
  timeout 600 python3 bench.py --workload iispt --steps 5 --warmup 2 --cpu-seconds 0 --net-batch $nb > $O/bench_iispt_$nb.json 2> $O/bench_iispt_$nb.err; python3 -c "
import json; j=json.loads(open('$O/bench_iispt_$nb.json').readline()); print($nb, j['ms_per_step'], j['stage_ms_per_step']['network'], j['roofline']['frac_executed'], j['roofline']['agreement_with_the_module'])"
done
timeout 600 python3 tools/net_check.py 8192 > $O/net_check.json 2> $O/net_check.err; cat $O/net_check.json | head -c 1500
