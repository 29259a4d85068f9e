#!/bin/bash
# round-4 third measurement call: refill rule (fixed idle-lane thresholds against the wasted-lane-steps rule), direct-pass and network tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call3
mkdir -p $O
cd $R
export AB_ARGS="--workload boxroom"
timeout 1500 tools/ab.sh default leaf3 l3idle8 l3idle12 l3idle16 l3idle24 l3w64 l3w96 l3w128 l3w192 l3w96g > $O/ab_room.txt 2>&1
export AB_ARGS=""
timeout 1200 tools/ab.sh default leaf3 l3idle16 l3idle24 l3w64 l3w96 l3w128 l3w192 l3w96g > $O/ab_killeroo.txt 2>&1
timeout 900 python3 -m pytest tests/test_iispt_direct.py tests/test_iispt_nn.py -m gpu -x -q -s > $O/tests_direct_nn.txt 2>&1
tail -5 $O/tests_direct_nn.txt
cat $O/ab_room.txt $O/ab_killeroo.txt
