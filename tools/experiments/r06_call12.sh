#!/bin/bash
# round-6 call 12: partial spheres and textures on spheres on the device — the new parity test, the whole GPU suite, the headline's time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call12
mkdir -p $O
cd $R
( time timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spheres" ) > $O/tests_spheres.txt 2>&1; tail -14 $O/tests_spheres.txt | head -11
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1; grep -n "passed\|failed" $O/tests.txt
( time timeout 900 python3 tools/fuzz_rooms.py 63000 150 ) > $O/fuzz_rooms.txt 2>&1; grep "mismatches" $O/fuzz_rooms.txt
( time timeout 600 python3 tools/fuzz_direct.py 5000 300 ) > $O/fuzz_direct.txt 2>&1; tail -4 $O/fuzz_direct.txt | head -2
timeout 600 python3 bench.py --sub-configs none --cpu-seconds 0 > $O/bench.json 2> $O/bench.err; python3 -c "
import json; j=json.loads(open('$O/bench.json').readline()); print(j['ms_per_step'], j['value'], j['kernel_ms_per_step_one_stream'])"
