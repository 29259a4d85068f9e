#!/bin/bash
# round-4 last call: the evidence set on the final tree and the IISPT frame figures
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
bash tools/evidence.sh r04d > gpurun_out/evidence_r04d.log 2>&1; tail -n 4 gpurun_out/evidence_r04d.log | cut -c1-300
O=$R/gpurun_out/r04_call25
mkdir -p $O
timeout 300 python3 tools/probe_bench.py 1920 1080 10 1 f32 > $O/iispt_frame_f32.json 2> $O/iispt_frame_f32.err; head -c 700 $O/iispt_frame_f32.json; echo
timeout 300 python3 tools/probe_bench.py 1920 1080 10 1 bf16 > $O/iispt_frame_bf16.json 2> $O/iispt_frame_bf16.err; head -c 700 $O/iispt_frame_bf16.json; echo
timeout 600 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests.txt | tail -n 2
