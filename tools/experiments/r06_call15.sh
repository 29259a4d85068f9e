#!/bin/bash
# round-6 call 15: does the IISPT direct pass hide under the network on a second stream?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call15
mkdir -p $O
cd $R
( time timeout 600 python3 tools/experiments/r06_two_stream_frame.py 6 ) > $O/two_stream.txt 2>&1; tail -8 $O/two_stream.txt
