#!/bin/bash
# round-6 call 6: uber's SpecularTransmission lobes on the device — the new parity test, the whole GPU suite (the material record changed),
# a randomised sweep of rooms that now includes the new materials
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call6
mkdir -p $O
cd $R
( time timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "uber_transmission or specular_materials or glass" ) > $O/tests_uber.txt 2>&1; tail -3 $O/tests_uber.txt
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $O/tests.txt 2>&1; tail -4 $O/tests.txt
( time timeout 900 python3 tools/fuzz_rooms.py 61000 400 ) > $O/fuzz_rooms.txt 2>&1; tail -3 $O/fuzz_rooms.txt; grep -c ubertrans $O/fuzz_rooms.txt; grep MISMATCH $O/fuzz_rooms.txt | head
timeout 600 python3 bench.py --sub-configs none --cpu-seconds 0 > $O/bench.json 2> $O/bench.err; python3 -c "
import json; j=json.loads(open('$O/bench.json').readline()); print(j['ms_per_step'], j['value'])"
timeout 600 python3 bench.py --workload boxroom-textured --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_textured.json 2> $O/bench_textured.err; python3 -c "
import json; j=json.loads(open('$O/bench_textured.json').readline()); print('textured room', j['ms_per_step'], j['kernel_ms_per_step_one_stream'])"
