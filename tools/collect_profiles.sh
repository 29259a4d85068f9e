#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   tools/collect_profiles.sh <tag> [bench.py arguments, e.g. --workload boxroom]   -> gpurun_out/prof_<tag>/{stats,fetch,write,sq,tcc,sq2,mem}/...
# (a tag containing "_room" marks the deep-tree workload: bench.py --workload boxroom loads profiles/*_room_pmc_*.json)
# Pass 1: --kernel-trace --stats of the bench command (--alone-steps 0: only the instrumented step, the warm-up and the
# timed steps launch kernels, so the plain builds' average durations are those of the two-stream schedule bench.py times). Passes 2-5: PMC counters in their own runs
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; SQ has 8 slots, TCC 4), kernel-trace only.
# Every profiler run is wrapped in `timeout` (a hung counter set once cost 20 GPU-minutes).
# tools/summarize_profiles.py <tag> then writes profiles/<tag>_{kernel_stats.csv,pmc_traffic.json,pmc_lanes.json}.
set -u
TAG=${1:-r03}
shift
EXTRA="$*"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ONE="$R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --other-steps 0 --alone-steps 0 --sub-configs none $EXTRA"   # (--sub-configs none: the default line's config-4 / config-5 blocks would put their kernels into these traces)
echo "$EXTRA" > "$O/bench_args.txt"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --other-steps 0 --alone-steps 0 --sub-configs none $EXTRA > "$O/stats.log" 2>&1
grep -h '"metric"' "$O/stats.log" | tail -1 > "$O/bench_under_profiler.json"
# the same trace with every kernel of the timed steps alone on the GPU (bench.py --schedule one-stream): the per-kernel averages the
# bench line's `roofline` / `roofline_all_kernels` are priced with
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_one_stream" -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --other-steps 0 --alone-steps 0 --sub-configs none --schedule one-stream $EXTRA > "$O/stats_one_stream.log" 2>&1
grep -h '"metric"' "$O/stats_one_stream.log" | tail -1 > "$O/bench_one_stream_under_profiler.json"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -- python3 $ONE > "$O/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -- python3 $ONE > "$O/write.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d "$O/sq" -- python3 $ONE > "$O/sq.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$O/tcc" -- python3 $ONE > "$O/tcc.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace --output-format csv -d "$O/sq2" -- python3 $ONE > "$O/sq2.log" 2>&1
timeout 600 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/mem" -- python3 $ONE > "$O/mem.log" 2>&1
find "$O" -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
tail -2 "$O"/*.log | cut -c1-300
