#!/bin/bash
# round-4 call 18: the IISPT runner's calls on the scene's scratch block: tests, the frame's stage times before / after
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call18
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_iispt_gather.py tests/test_iispt_direct.py tests/test_iispt_nn.py -x -q -m gpu > $O/tests.txt 2>&1; tail -n 4 $O/tests.txt
ls tests | grep -i "iispt\|probe" > $O/testfiles.txt
IILE_IISPT_TIMERS=1 timeout 200 python3 tools/probe_bench.py 1920 1080 10 1 f32 > $O/frame_f32_timers.json 2> $O/frame_f32_timers.err; tail -c 1500 $O/frame_f32_timers.json
timeout 200 python3 tools/probe_bench.py 1920 1080 10 1 f32 > $O/frame_f32.json 2> $O/frame_f32.err; tail -c 900 $O/frame_f32.json
timeout 200 python3 tools/probe_bench.py 1920 1080 10 1 bf16 > $O/frame_bf16.json 2> $O/frame_bf16.err; tail -c 900 $O/frame_bf16.json
