#!/bin/bash
# round-6 call 17: anisotropic roughness (uber, glass) — bitwise tests, fuzz with the new room mode, headline and room timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call17
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anisotropic or rough_glass or uber_transmission or specular_materials or bsdf" ) > $O/tests.txt 2>&1; tail -14 $O/tests.txt | head -11
( time timeout 1200 python3 tools/fuzz_rooms.py 63000 168 iispt ) > $O/fuzz.txt 2>&1; tail -4 $O/fuzz.txt; grep -c aniso $O/fuzz.txt; grep "MISMATCH" $O/fuzz.txt | head
timeout 600 python3 bench.py --steps 10 --warmup 3 --cpu-seconds 0 --sub-cpu-seconds 0 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
j = json.loads(open('$O/bench.json').readline())
print(j['ms_per_step'], j['kernel_ms_per_step_one_stream'])
for k, b in j['configs'].items():
    print(k, b['ms_per_step'], b['value'])
PY
