#!/bin/bash
# round-4 second measurement call: load-width calibration, occupancy / stack-depth / vote variants on top of leaf3, vote statistics
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call2
mkdir -p $O
cd $R
timeout 400 tools/_build/vmem_calib 2000 > $O/vmem_calib.json 2> $O/vmem_calib.err
export AB_ARGS="--workload boxroom"
timeout 1500 tools/ab.sh default leaf3 l3w5 l3w4 l3idle16 l3v11 l3v32 > $O/ab_room.txt 2>&1
export AB_ARGS=""
timeout 900 tools/ab.sh default leaf3 l3w5 l3v11 l3v32 > $O/ab_killeroo.txt 2>&1
IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_iterstats.so timeout 300 python3 tools/trav_stamps.py boxroom iterstats > $O/iterstats_room.json 2> $O/iterstats_room.err
IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_iterstats.so timeout 300 python3 tools/trav_stamps.py killeroo iterstats > $O/iterstats_killeroo.json 2> $O/iterstats_killeroo.err
cat $O/ab_room.txt $O/ab_killeroo.txt $O/iterstats_room.json
