#!/bin/bash
# round-4 call 13: the whole GPU suite + smoke on the final tree; then larger traversal blocks (768 threads: one LDS copy of the tree's top
# for twelve wavefronts) with 85 top records or a 12th stack level, against the 256-thread default
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call13
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
for v in b768t85 b768L12; do
  IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests_$v.txt 2>&1; tail -2 $O/tests_$v.txt
done
AB_ARGS="--workload boxroom --steps 4 --warmup 1 --alone-steps 2" bash tools/ab.sh default b768t21 b768t85 b768L12 > $O/ab_room.txt 2>&1; cat $O/ab_room.txt
AB_ARGS="--steps 10 --warmup 2 --alone-steps 3" bash tools/ab.sh default b768t21 b768t85 b768L12 > $O/ab_killeroo.txt 2>&1; cat $O/ab_killeroo.txt
