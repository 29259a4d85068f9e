#!/bin/bash
# round-4 fourth measurement call: the ramp refill rule, the split axes in the refs; parity of the combination
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call4
mkdir -p $O
cd $R
export AB_ARGS="--workload boxroom"
timeout 1500 tools/ab.sh leaf3 l3idle12 l3ramp l3ramp12 l3ramp40 l3ramp4 l3axes l3axramp > $O/ab_room.txt 2>&1
export AB_ARGS=""
timeout 1200 tools/ab.sh leaf3 l3idle24 l3ramp l3ramp12 l3ramp40 l3ramp4 l3axes l3axramp > $O/ab_killeroo.txt 2>&1
IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_l3axramp.so timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bvh_build.py -m gpu -x -q > $O/parity_l3axramp.txt 2>&1
tail -3 $O/parity_l3axramp.txt
cat $O/ab_room.txt $O/ab_killeroo.txt
