#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call8
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_iispt_direct.py tests/test_iispt_gather.py tests/test_iispt_nn.py -m gpu -x -q > $O/tests.txt 2>&1
tail -15 $O/tests.txt
