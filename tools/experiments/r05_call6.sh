#!/bin/bash
# round-5 call 6: the one-stream kernel traces again with the warm-up step on the same schedule (the first set mixed one two-stream
# warm-up into the averages), the IISPT frame tests with the one-launch film update, the frame's bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_call6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in killeroo boxroom; do
  rm -rf $O/one_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one_$w -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --other-steps 0 --alone-steps 0 --schedule one-stream --workload $w > $O/one_$w.log 2>&1
  find $O/one_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/one_stream_${w}_kernel_stats.csv
  grep -h '"metric"' $O/one_$w.log | tail -1 > $O/one_stream_${w}_bench.json
done
cd $R
timeout 900 python -m pytest tests/test_iispt_gather.py tests/test_iispt_nn.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python bench.py --workload iispt --steps 5 --warmup 2 --cpu-seconds 0 > $O/bench_iispt.txt 2>&1; tail -1 $O/bench_iispt.txt | cut -c1-1400
