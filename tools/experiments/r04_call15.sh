#!/bin/bash
# round-4 call 15: buildUpperSAH on the device (k_upper_sah): the builder's tests, the randomised builder sweep, build times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call15
mkdir -p $O
cd $R
timeout 40 python3 tools/experiments/r04_hand_case.py > $O/hand.txt 2>&1; tail -n 3 $O/hand.txt
grep -q "hand case ok" $O/hand.txt || exit 1
timeout 200 python -m pytest tests/test_gpu_bvh_build.py -x -q -m gpu > $O/tests_bvh.txt 2>&1; tail -n 5 $O/tests_bvh.txt
timeout 200 python3 tools/fuzz_bvh.py > $O/fuzz_bvh.txt 2>&1; tail -n 3 $O/fuzz_bvh.txt
timeout 200 python3 tools/bvh_build_bench.py > $O/bvh_build_bench.json 2> $O/bvh_bench.err; tail -n 3 $O/bvh_bench.err; head -c 3000 $O/bvh_build_bench.json
