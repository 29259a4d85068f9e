#!/bin/bash
# round-5 call 3: the network's GPU tests on the HIP path, bench.py --workload iispt (config 5's contract line), its steady-state
# kernel trace (warm-up frames first, three timed frames), and the IISPT frame tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_call3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_iispt_nn.py tests/test_iispt_gather.py tests/test_abi.py -x -q -m gpu -s > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 600 python bench.py --workload iispt --steps 5 --warmup 2 > $O/bench_iispt.txt 2>&1; tail -1 $O/bench_iispt.txt | cut -c1-3000
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload iispt --steps 3 --warmup 2 --cpu-seconds 0 > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -40 $O/kernel_stats.csv | cut -c1-200
