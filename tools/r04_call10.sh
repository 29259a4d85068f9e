#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call10
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_iispt_direct.py tests/test_gpu_bvh_build.py -m gpu -x -q > $O/tests.txt 2>&1
tail -15 $O/tests.txt
timeout 300 python3 tools/bvh_build_bench.py > $O/bvh_build_bench.json 2> $O/bvh_build_bench.err; tail -5 $O/bvh_build_bench.json | cut -c1-400
