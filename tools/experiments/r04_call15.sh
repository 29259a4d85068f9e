#!/bin/bash
# round-4 call 15: buildUpperSAH on the device (k_upper_sah): the builder's tests, the randomised builder sweep, build times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call15
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bvh_build.py -x -q -m gpu > $O/tests_bvh.txt 2>&1; tail -n 15 $O/tests_bvh.txt
timeout 600 python3 tools/fuzz_bvh.py > $O/fuzz_bvh.txt 2>&1; tail -n 3 $O/fuzz_bvh.txt
timeout 600 python3 tools/bvh_build_bench.py $O/bvh_build_bench.json > $O/bvh_bench.txt 2>&1; tail -n 5 $O/bvh_bench.txt; cat $O/bvh_build_bench.json | head -c 3000
