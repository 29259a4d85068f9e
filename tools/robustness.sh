#!/bin/bash
# robustness + config-5 figures (rounds 4, 5): randomised rooms / gather / builder sweeps, the 1080p x 1024 spp frame against the oracle
# (sixteen passes: the device film finish across passes), the IISPT frame in fp32 and bf16
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/robustness
mkdir -p $O
cd $R
timeout 1500 python3 tools/fuzz_rooms.py 4000 120 > $O/fuzz_rooms.txt 2>&1; tail -2 $O/fuzz_rooms.txt
timeout 900 python3 tools/fuzz_direct.py 300 72 > $O/fuzz_direct.txt 2>&1; tail -2 $O/fuzz_direct.txt
timeout 600 python3 tools/fuzz_gather.py > $O/fuzz_gather.txt 2>&1; tail -2 $O/fuzz_gather.txt
timeout 600 python3 tools/fuzz_bvh.py > $O/fuzz_bvh.txt 2>&1; tail -2 $O/fuzz_bvh.txt
timeout 1800 python3 tools/full_frame_parity.py $O/full_frame_parity_1024spp.json killeroo 1024 > $O/full_frame_1024.txt 2>&1; tail -3 $O/full_frame_1024.txt
timeout 1200 python3 tools/full_frame_parity.py $O/full_frame_parity_boxroom_textured.json boxroom-textured 16 > $O/full_frame_tex.txt 2>&1; tail -3 $O/full_frame_tex.txt
timeout 900 python3 tools/probe_bench.py 1920 1080 10 1 hip > $O/iispt_frame_hip.json 2> $O/iispt_frame_hip.err; cat $O/iispt_frame_hip.json | cut -c1-600
