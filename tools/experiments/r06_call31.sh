#!/bin/bash
# round-6 call 31: the C++ IISPT host over several GPUs (--gpurank): world-1 communicator branch, shares against the Python shares
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call31
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_iispt_host.py tests/test_device_gang.py tests/test_gpu_configs.py -m gpu -x -q -k "iispt or shard or gang or deadline or dist" ) > $O/tests.txt 2>&1; tail -30 $O/tests.txt | head -26
