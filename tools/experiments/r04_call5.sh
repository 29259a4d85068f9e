#!/bin/bash
# round-4 fifth measurement call: adaptive refill batch; infinite lights in the direct pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call5
mkdir -p $O
cd $R
export AB_ARGS="--workload boxroom"
timeout 1500 tools/ab.sh ax32 ax12 ax12g24 ax12g40 axad axad3 axad1 axadg > $O/ab_room.txt 2>&1
export AB_ARGS=""
timeout 1200 tools/ab.sh ax32 ax12 axad axad3 axad1 axadg > $O/ab_killeroo.txt 2>&1
export AB_ARGS="--workload boxroom-textured"
timeout 900 tools/ab.sh ax32 axad axad3 > $O/ab_roomtex.txt 2>&1
timeout 900 python3 -m pytest tests/test_iispt_direct.py -m gpu -x -q > $O/tests_direct.txt 2>&1
tail -5 $O/tests_direct.txt
cat $O/ab_room.txt $O/ab_killeroo.txt $O/ab_roomtex.txt
