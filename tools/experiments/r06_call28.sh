#!/bin/bash
# round-6 call 28: float images for uroughness / vroughness / roughness on uber — bitwise tests, textured tests, a fuzz pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call28
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_iispt_direct.py -m gpu -x -q -k "anisotropic or textur or uber or rough or bsdf or alpha" ) > $O/tests.txt 2>&1; tail -12 $O/tests.txt | head -8
( time timeout 900 python3 tools/fuzz_rooms.py 80000 420 iispt ) > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt | head -1; grep -c "OK/OK/OK" $O/fuzz.txt; grep MISMATCH $O/fuzz.txt | head -3
