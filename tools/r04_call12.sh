#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call12
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_iispt_direct.py -m gpu -x -q > $O/tests.txt 2>&1
tail -6 $O/tests.txt
timeout 600 python3 bench.py --workload boxroom-textured --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_boxroom_textured.json 2> $O/bench_boxroom_textured.err
python3 -c "
import json
d=json.loads(open('$O/bench_boxroom_textured.json').readline()); print(d['ms_per_step'], d['kernel_ms_per_step_one_stream'])"
timeout 600 python3 bench.py --steps 10 --warmup 2 --cpu-seconds 0 --other-steps 0 > $O/bench.json 2> $O/bench.err
python3 -c "
import json
d=json.loads(open('$O/bench.json').readline()); print(d['ms_per_step'], d['kernel_ms_per_step_one_stream'])"
