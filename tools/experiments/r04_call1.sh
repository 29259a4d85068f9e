#!/bin/bash
# round-4 first measurement call: vector-memory calibration, A/B of the traversal variants on the room and killeroo,
# record-order scramble, vote statistics
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call1
mkdir -p $O
cd $R
timeout 300 tools/_build/vmem_calib 2000 > $O/vmem_calib.json 2> $O/vmem_calib.err
export AB_ARGS="--workload boxroom"
timeout 1200 tools/ab.sh default leaf3 int2 int3 xcd leaf3xcd > $O/ab_room.txt 2>&1
IILE_RECORD_ORDER=scramble timeout 300 tools/ab.sh default > $O/ab_room_scramble.txt 2>&1
unset AB_ARGS
export AB_ARGS=""
timeout 900 tools/ab.sh default leaf3 int2 xcd leaf3xcd > $O/ab_killeroo.txt 2>&1
IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_iterstats.so timeout 300 python3 tools/trav_stamps.py boxroom iterstats > $O/iterstats_room.json 2> $O/iterstats_room.err
IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_iterstats.so timeout 300 python3 tools/trav_stamps.py killeroo iterstats > $O/iterstats_killeroo.json 2> $O/iterstats_killeroo.err
IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_leaf3xcd.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/parity_leaf3xcd.txt 2>&1
tail -3 $O/parity_leaf3xcd.txt
cat $O/ab_room.txt
