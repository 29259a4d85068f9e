#!/bin/bash
# round-4 call 20: the whole GPU suite and smoke on the final tree, the robustness sweeps, the 1-rank torchrun line with the fixed
# counter applicability
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r04_call20
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/gpu_tests.txt | tail -n 3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 2 $O/smoke.txt
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 3 --warmup 1 --scaling strong --cpu-seconds 0 --other-steps 0 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; head -c 400 $O/bench_torchrun1.json; echo
bash tools/robustness.sh > $O/robustness.log 2>&1; tail -n 12 $O/robustness.log | cut -c1-600
