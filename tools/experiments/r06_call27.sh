#!/bin/bash
# round-6 call 27: pixelbounds through the C++ host
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call27
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pixelbounds" ) > $O/tests.txt 2>&1; tail -12 $O/tests.txt | head -8
