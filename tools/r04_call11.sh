#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call11
mkdir -p $O
cd $R
timeout 2400 python3 tools/full_frame_parity.py $O/full_frame_parity_boxroom_256spp.json boxroom 256 > $O/full_frame_room.txt 2>&1; tail -3 $O/full_frame_room.txt | cut -c1-900
