#!/bin/bash
# round-6 call 20: k_direct_shade at 3 / 4 waves per SIMD (spilling 66 / 103 dwords) against the shipped 2 (no spills)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call20
mkdir -p $O
cd $R
for rep in 1 2; do
for v in default dshade3 dshade4; do
  if [ "$v" = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  echo "$v: $(timeout 300 python3 tools/prof_direct.py 16 6 2>&1 | tail -1)"
done; done | tee $O/ab_direct_shade_waves.txt
