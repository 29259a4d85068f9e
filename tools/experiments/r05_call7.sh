#!/bin/bash
# round-5 call 7: rehearse the build ON the GPU box (every driver run so far used the libraries shipped from the build container):
# remove the shipped libraries, let __graft_entry__.build() compile everything with the box's hipcc, run smoke, a parity subset
# and the headline bench line (its `built.compiled_on_this_box` must read true)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_call7
mkdir -p $O
cd $R
rm -f pbrt-v3-iile_amd/lib/*.so pbrt-v3-iile_amd/lib/iile_pbrt pbrt-v3-iile_amd/csrc/device/*.o pbrt-v3-iile_amd/csrc/host/*.o oracle/_build/liboracle.so
hipcc --version 2>&1 | head -2 > $O/hipcc_on_box.txt; cat /opt/rocm/.info/version >> $O/hipcc_on_box.txt 2>/dev/null
( time timeout 1500 python bench.py --steps 10 --warmup 2 ) > $O/bench_built_on_box.txt 2> $O/bench_built_on_box.err; tail -1 $O/bench_built_on_box.txt | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(json.dumps({'value': j['value'], 'ms_per_step': j['ms_per_step'], 'built': j['built']}))"
tail -4 $O/bench_built_on_box.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_iispt_nn.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
