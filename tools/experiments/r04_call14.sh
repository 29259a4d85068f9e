#!/bin/bash
# round-4 call 14: what k_shadow's one scattered access (the 16-byte L[pid] read-modify-write) costs: a timing-only build without it
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call14
mkdir -p $O
cd $R
for rep in 1 2; do
for v in default shadow_noL; do
  if [ "$v" = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  echo "$v killeroo: $(timeout 300 python3 tools/prof_render.py 1920 1080 64 4 2>&1 | tail -n 1)" >> $O/shadow_noL.txt
done; done
unset IILE_GPU_LIB
cat $O/shadow_noL.txt
bash tools/robustness.sh > $O/robustness.log 2>&1; tail -n 12 $O/robustness.log
