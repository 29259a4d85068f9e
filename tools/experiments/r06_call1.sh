#!/bin/bash
# round-6 call 1: the tree after the housekeeping (torch module out of the package, one stream through the IISPT calls, deadline
# communicator, gang vote) — whole GPU suite, the bench line with its new sub-blocks, and the fp16-subnormal question of the matrix pipe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call1
mkdir -p $O
cd $R
tools/_build/mfma_f16_denorm > $O/mfma_f16_denorm.jsonl 2>&1; tail -1 $O/mfma_f16_denorm.jsonl
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1; tail -4 $O/tests.txt
( time timeout 900 python3 bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python3 - <<PY
import json
try:
    j = json.loads(open('$O/bench_default.json').readline())
    print({k: j[k] for k in ('value', 'ms_per_step')}, j['roofline']['frac'])
    for k, b in j.get('configs', {}).items():
        print(k, b['ms_per_step'], b['value'], b.get('wall_seconds_of_this_block'), b['roofline'].get('frac'), b['roofline'].get('frac_executed'), b.get('cpu_baseline', {}).get('value'))
        if 'stage_ms_per_step' in b: print(b['stage_ms_per_step'], b['roofline']['agreement_with_the_module'])
except Exception as e:
    print('bench parse failed', e)
PY
timeout 600 python3 bench.py --workload iispt --steps 5 --warmup 2 --cpu-seconds 0 > $O/bench_iispt.json 2> $O/bench_iispt.err; python3 -c "
import json; j=json.loads(open('$O/bench_iispt.json').readline()); print(j['ms_per_step'], j['stage_ms_per_step'])"
