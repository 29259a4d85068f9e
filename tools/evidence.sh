#!/bin/bash
# round-4 evidence: rocprofv3 stats + PMC passes of the bench command (killeroo, room), the summaries, the bench lines that read them,
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
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 3 --warmup 1 --scaling strong --cpu-seconds 0 --other-steps 0 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
ls -la $O | head -40
head -c 600 $O/bench.json; echo; head -c 300 $O/bench_boxroom.json; echo; for f in bench bench_boxroom bench_torchrun1; do tail -n 2 $O/$f.err; done
