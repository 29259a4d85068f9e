#!/bin/bash
# round-6 call 18: uber "opacity" as an image texture — bitwise tests, the textured tests again
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call18
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_iispt_direct.py -m gpu -x -q -k "uber_transmission or textur or anisotropic or alpha" ) > $O/tests.txt 2>&1; tail -14 $O/tests.txt | head -11
( time timeout 1200 python3 tools/fuzz_rooms.py 64000 252 iispt ) > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt; grep -c "OK/OK/OK" $O/fuzz.txt; grep "MISMATCH" $O/fuzz.txt | head
