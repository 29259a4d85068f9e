#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write}/...
# Pass 1: --kernel-trace --stats of the bench command. Passes 2/3: PMC FETCH_SIZE and
# WRITE_SIZE in their own runs (they do not fit one pass on gfx950), kernel-trace only.
# Every profiler run is wrapped in `timeout` (a hung counter set once cost 20 GPU-minutes).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $BENCH > "$O/stats.log" 2>&1
grep -h '"metric"' "$O/stats.log" | tail -1 > "$O/bench_under_profiler.json"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > "$O/fetch.log" 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 > "$O/write.log" 2>&1
find "$O" -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
