#!/bin/bash
# round-6 call 24: "pixelbounds" — bitwise tests, then the film / configs tests again (path_pixel and the exact finish were touched), headline timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call24
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pixelbounds" ) > $O/tests_pb.txt 2>&1; tail -30 $O/tests_pb.txt | head -26
( time timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_sobol.py -m gpu -x -q ) > $O/tests.txt 2>&1; tail -6 $O/tests.txt | head -3
timeout 600 python3 bench.py --steps 10 --warmup 3 --cpu-seconds 0 --sub-cpu-seconds 0 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
j = json.loads(open('$O/bench.json').readline())
print(j['ms_per_step'], j['kernel_ms_per_step_one_stream'])
for k, b in j['configs'].items():
    print(k, b['ms_per_step'], b['value'])
PY
