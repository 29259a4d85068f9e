#!/bin/bash
# tools/ab.sh NAME...   (on the GPU box) bench the default library and each named variant
R=${GRAFT_REPO_ROOT:-/root/repo}
show='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], d["ms_per_step"], {k[3:]:round(v,2) for k,v in d["kernel_ms_per_step_rank0"].items()})'
timeout 200 python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "$show" base
# NAME or NAME@VAR=value (environment for that run only)
for spec in "$@"; do
  n=${spec%%@*}; e=""
  if [ "$n" != "$spec" ]; then e=${spec#*@}; fi
  env IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$n.so $e timeout 200 python $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | python -c "$show" $spec
done
