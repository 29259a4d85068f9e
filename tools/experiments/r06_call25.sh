#!/bin/bash
# round-6 call 25: pixelbounds ignored by the IISPT entry points; direct-pass tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call25
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_iispt_direct.py tests/test_iispt_gather.py tests/test_iispt_host.py -m gpu -x -q -k "pixelbounds or direct or gather or host or cli" ) > $O/tests.txt 2>&1; tail -6 $O/tests.txt | head -4
