#!/bin/bash
# round-6 call 21: robustness campaign on the final tree — random rooms through every stage, the direct pass, the gather, the builder,
# the 1080p x 1024 spp frame and the textured room against the oracle
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call21
mkdir -p $O
cd $R
( time timeout 1500 python3 tools/fuzz_rooms.py 70000 1260 iispt ) > $O/fuzz_rooms.txt 2>&1; tail -5 $O/fuzz_rooms.txt | head -2; grep -c "OK/OK/OK" $O/fuzz_rooms.txt; grep MISMATCH $O/fuzz_rooms.txt | head -5
( time timeout 900 python3 tools/fuzz_direct.py 7000 252 ) > $O/fuzz_direct.txt 2>&1; tail -5 $O/fuzz_direct.txt | head -2
( time timeout 600 python3 tools/fuzz_gather.py 100 60 ) > $O/fuzz_gather.txt 2>&1; tail -5 $O/fuzz_gather.txt | head -2
( time timeout 600 python3 tools/fuzz_bvh.py 100 40 ) > $O/fuzz_bvh.txt 2>&1; tail -5 $O/fuzz_bvh.txt | head -2
( time timeout 1800 python3 tools/full_frame_parity.py $O/full_frame_parity_1024spp.json killeroo 1024 ) > $O/full_frame_1024.txt 2>&1; tail -6 $O/full_frame_1024.txt | head -3
( time timeout 1200 python3 tools/full_frame_parity.py $O/full_frame_parity_boxroom_textured.json boxroom-textured 16 ) > $O/full_frame_tex.txt 2>&1; tail -6 $O/full_frame_tex.txt | head -3
