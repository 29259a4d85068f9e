#!/bin/bash
# round-4 sixth call: the new defaults (leaf records together, axes in refs, adaptive refill, device film finish) through the whole
# GPU test suite; A/B of the adaptive rule's start / rate against the fixed batches
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call6
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/tests_gpu.txt 2>&1
tail -8 $O/tests_gpu.txt
export AB_ARGS="--workload boxroom"
timeout 1200 tools/ab.sh default adi3 adi4 ad4 fix12 fix32 > $O/ab_room.txt 2>&1
export AB_ARGS=""
timeout 1200 tools/ab.sh default adi3 adi4 ad4 fix12 fix32 > $O/ab_killeroo.txt 2>&1
cat $O/ab_room.txt $O/ab_killeroo.txt
