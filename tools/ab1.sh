#!/bin/bash
# like ab.sh, but prints the one-stream per-kernel times (every kernel alone on the GPU) beside the step time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --other-steps 0 --alone-steps 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); k=j['kernel_ms_per_step_one_stream']
print('%-12s %7.3f ms/step | alone: ext %.2f shade %.2f shadow %.2f mis %.2f film %.2f' % ('$v', j['ms_per_step'], k['ms_extend'], k['ms_shade'], k['ms_shadow'], k['ms_mis'], k['ms_film']))"
done; done
