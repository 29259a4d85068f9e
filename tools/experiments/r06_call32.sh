#!/bin/bash
# round-6 call 32: k_direct_shade working out two light samples side by side (-DIILE_DIRECT_SHADE_PAIRS) against the shipped kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call32
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_iispt_direct.py -m gpu -x -q ) > $O/tests_default.txt 2>&1; grep -E "passed|failed" $O/tests_default.txt
( export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_dpairs.so; time timeout 900 python3 -m pytest tests/test_iispt_direct.py -m gpu -x -q ) > $O/tests_pairs.txt 2>&1; grep -E "passed|failed" $O/tests_pairs.txt
for rep in 1 2 3; do
for v in default dpairs; do
  if [ "$v" = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  echo "$v: $(timeout 300 python3 tools/prof_direct.py 16 6 2>&1 | tail -1)"
done; done | tee $O/ab_direct_shade_pairs.txt
