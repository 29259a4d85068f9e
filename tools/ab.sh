#!/bin/bash
# A/B of libiile_gpu builds on the GPU box: tools/ab.sh [variant names under pbrt-v3-iile_amd/lib/variants, "default" = the in-tree build]
# prints ms/step and per-kernel ms (one-stream steps) of `bench.py --steps 10` for each, twice (interleaved, to see the run-to-run spread)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --other-steps 0 $AB_ARGS 2>/tmp/ab_err.log | python3 -c "
import json,sys
l=sys.stdin.readline()
if not l.strip():
    print('%-12s FAILED: %s' % ('$v', open('/tmp/ab_err.log').read()[-300:].replace(chr(10),' | ')))
else:
    j=json.loads(l); k=j['kernel_ms_per_step_one_stream']
    print('%-12s %7.3f ms/step  one-stream: ext %.2f shade %.2f shadow %.2f mis %.2f film %.2f' % ('$v', j['ms_per_step'], k['ms_extend'], k['ms_shade'], k['ms_shadow'], k['ms_mis'], k['ms_film']))"
done; done
