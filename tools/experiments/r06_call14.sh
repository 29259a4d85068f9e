#!/bin/bash
# round-6 call 14: uber transmission through the IISPT stages (direct pass with two specular-transmission lobes), fuzz over the stages
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call14
mkdir -p $O
cd $R
( time timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rough_glass or uber_transmission or glass or direct" ) > $O/tests.txt 2>&1; tail -14 $O/tests.txt | head -11
( time timeout 1200 python3 tools/fuzz_rooms.py 62000 140 iispt ) > $O/fuzz_iispt.txt 2>&1; tail -4 $O/fuzz_iispt.txt; grep -c "iispt direct" $O/fuzz_iispt.txt; grep -c refused $O/fuzz_iispt.txt; grep "MISMATCH" $O/fuzz_iispt.txt | head
