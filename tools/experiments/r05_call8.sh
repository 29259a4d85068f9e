#!/bin/bash
# round-5 call 8: larger randomised sweeps on the final tree (new seeds): rooms, direct pass, gather, BVH builder against the oracle
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_call8
mkdir -p $O
cd $R
( time timeout 1300 python3 tools/fuzz_rooms.py 5000 400 ) > $O/fuzz_rooms.txt 2>&1; tail -5 $O/fuzz_rooms.txt
( time timeout 700 python3 tools/fuzz_direct.py 700 200 ) > $O/fuzz_direct.txt 2>&1; tail -5 $O/fuzz_direct.txt
( time timeout 400 python3 tools/fuzz_gather.py 100 160 ) > $O/fuzz_gather.txt 2>&1; tail -5 $O/fuzz_gather.txt
( time timeout 400 python3 tools/fuzz_bvh.py 100 100 ) > $O/fuzz_bvh.txt 2>&1; tail -5 $O/fuzz_bvh.txt
