#!/bin/bash
# round-5 call 9: the final tree — headline bench line, a big randomised sweep (new seeds), smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_call9
mkdir -p $O
cd $R
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; j=json.loads(open('$O/bench_default.json').readline()); print(json.dumps({k: j[k] for k in ('value','ms_per_step','n_gpus','steps')}), j['roofline']['frac'], j['cpu_baseline']['value'])"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python3 tools/fuzz_rooms.py 20000 3000 ) > $O/fuzz_rooms.txt 2>&1; tail -5 $O/fuzz_rooms.txt | head -2
( time timeout 600 python3 tools/fuzz_direct.py 3000 1500 ) > $O/fuzz_direct.txt 2>&1; tail -5 $O/fuzz_direct.txt | head -2
( time timeout 300 python3 tools/fuzz_gather.py 1000 2000 ) > $O/fuzz_gather.txt 2>&1; tail -5 $O/fuzz_gather.txt | head -2
( time timeout 600 python3 tools/fuzz_bvh.py 1000 600 ) > $O/fuzz_bvh.txt 2>&1; tail -5 $O/fuzz_bvh.txt | head -2
