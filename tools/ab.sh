#!/bin/bash
# A/B of libiile_gpu builds on the GPU box: tools/ab.sh [variant names under pbrt-v3-iile_amd/lib/variants, "default" = the in-tree build]
# prints ms/step and per-kernel ms of `bench.py --steps 10` for each, twice (interleaved, to see the run-to-run spread)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --other-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); k=j['kernel_ms_per_step_rank0']
print('%-12s %7.3f ms/step  ext %.2f shade %.2f shadow %.2f mis %.2f film %.2f' % ('$v', j['ms_per_step'], k['ms_extend'], k['ms_shade'], k['ms_shadow'], k['ms_mis'], k['ms_film']))"
done; done
