#!/bin/bash
# round-6 call 11: rough glass on the device — its parity test, the whole GPU suite, a randomised sweep that includes the new rooms, the textured room's time
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call11
mkdir -p $O
cd $R
( time timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rough_glass or uber_transmission or glass or bsdf" ) > $O/tests_glass.txt 2>&1; tail -12 $O/tests_glass.txt | head -9
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1; grep -n "passed\|failed" $O/tests.txt
( time timeout 900 python3 tools/fuzz_rooms.py 62000 200 ) > $O/fuzz_rooms.txt 2>&1; tail -3 $O/fuzz_rooms.txt | head -1; grep -c roughglass $O/fuzz_rooms.txt; grep MISMATCH $O/fuzz_rooms.txt | head
timeout 600 python3 bench.py --workload boxroom-textured --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_textured.json 2> $O/bench_textured.err; python3 -c "
import json; j=json.loads(open('$O/bench_textured.json').readline()); print('textured room', j['ms_per_step'], j['kernel_ms_per_step_one_stream'])"
timeout 600 python3 bench.py --sub-configs none --cpu-seconds 0 > $O/bench.json 2> $O/bench.err; python3 -c "
import json; j=json.loads(open('$O/bench.json').readline()); print(j['ms_per_step'], j['value'])"
