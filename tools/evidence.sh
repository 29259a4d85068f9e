#!/bin/bash
# evidence of a round (r04d, r05c ...): rocprofv3 stats + PMC passes of the bench command (killeroo, room), the summaries, the bench lines that read them,
# the vmem calibration's own counters, the 1-rank torchrun line, the room at BASELINE's 256 spp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04d}
O=$R/gpurun_out/${TAG}_evidence
mkdir -p $O
cd $R
bash tools/collect_profiles.sh $TAG > $O/collect_killeroo.log 2>&1
python3 tools/summarize_profiles.py $TAG > $O/summarize_killeroo.log 2>&1
bash tools/collect_profiles.sh ${TAG}_room --workload boxroom > $O/collect_room.log 2>&1
python3 tools/summarize_profiles.py ${TAG}_room > $O/summarize_room.log 2>&1
cp profiles/${TAG}_* $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum --kernel-trace --output-format csv -d $O/vmem_pmc -- $R/tools/_build/vmem_calib 400 > $O/vmem_calib_under_pmc.json 2> $O/vmem_pmc.err
cd $R
timeout 600 python3 bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 600 python3 bench.py --workload boxroom --steps 5 --warmup 1 > $O/bench_boxroom.json 2> $O/bench_boxroom.err
timeout 600 python3 bench.py --workload boxroom-textured --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_boxroom_textured.json 2> $O/bench_boxroom_textured.err
timeout 600 python3 bench.py --sampler sobol --steps 10 --warmup 2 --cpu-seconds 0 --other-steps 0 > $O/bench_sobol.json 2> $O/bench_sobol.err
timeout 900 python3 bench.py --workload boxroom --spp 256 --steps 2 --warmup 1 --cpu-seconds 0 --alone-steps 1 > $O/bench_boxroom_256spp.json 2> $O/bench_boxroom_256spp.err
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 3 --warmup 1 --scaling strong --cpu-seconds 0 --other-steps 0 --sub-configs none > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
# BASELINE config 5: the IISPT frame — the contract line, its steady-state kernel trace (two warm-up frames first), the
# network kernels' matrix-pipe counters, and ten fresh processes of the same line (is any process at half speed?)
timeout 600 python3 bench.py --workload iispt --steps 5 --warmup 2 > $O/bench_iispt.json 2> $O/bench_iispt.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/iispt_stats -- python3 $R/bench.py --workload iispt --steps 3 --warmup 2 --cpu-seconds 0 > $O/iispt_stats.log 2>&1 )
find $O/iispt_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_iispt_kernel_stats.csv
bash tools/net_pmc.sh default > $O/net_pmc.log 2>&1; cp gpurun_out/net_pmc/default.txt $O/${TAG}_net_pmc.txt
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python3 bench.py --workload iispt --steps 3 --warmup 2 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(json.dumps({'run': $i, 'ms_per_step': j['ms_per_step'], 'network_ms': j['stage_ms_per_step']['network'], 'probes_per_s': j['value']}))"; done > $O/${TAG}_iispt_ten_processes.jsonl
ls -la $O | head -60
head -c 600 $O/bench.json; echo; head -c 300 $O/bench_boxroom.json; echo; for f in bench bench_boxroom bench_torchrun1; do tail -n 2 $O/$f.err; done
