#!/bin/bash
# round-5 call 5: leaf items (pairs of triangles sharing an edge: four loads for two tests) — the traversal parity tests, then A/B
# against the build before (variants/libiile_gpu_prev.so) on both workloads
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_call5
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bvh_build.py -x -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt
AB_ARGS="--workload boxroom --steps 4 --warmup 1 --alone-steps 2" bash tools/ab.sh prev default > $O/ab_room.txt 2>&1; cat $O/ab_room.txt
AB_ARGS="--steps 10 --warmup 2 --alone-steps 2" bash tools/ab.sh prev default > $O/ab_killeroo.txt 2>&1; cat $O/ab_killeroo.txt
