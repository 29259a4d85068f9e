#!/bin/bash
# round-6 call 13: rough glass through the IISPT stages (direct pass, hemi points, gather)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call13
mkdir -p $O
cd $R
( time timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rough_glass or uber_transmission" ) > $O/tests.txt 2>&1; tail -14 $O/tests.txt | head -11
