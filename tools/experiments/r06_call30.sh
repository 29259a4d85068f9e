#!/bin/bash
# round-6 call 30: the IISPT frame in shards — emulation test, the bench line at N = 1 through torchrun, the predicted scaling
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call30
mkdir -p $O
cd $R
( time timeout 900 python3 -m pytest tests/test_iispt_host.py -m gpu -x -q ) > $O/tests.txt 2>&1; tail -8 $O/tests.txt | head -5
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --workload iispt --gpus 1 --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_iispt_torchrun1.json 2> $O/bench_iispt_torchrun1.err; tail -1 $O/bench_iispt_torchrun1.json | cut -c1-300; tail -2 $O/bench_iispt_torchrun1.err
timeout 600 python3 bench.py --workload iispt --steps 3 --warmup 1 --cpu-seconds 0 2>/dev/null | cut -c1-200
( time timeout 900 python3 tools/iispt_shard_probe.py $O/iispt_shard_probe.json ) > $O/shard_probe.txt 2>&1; tail -40 $O/shard_probe.txt
