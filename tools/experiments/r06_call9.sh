#!/bin/bash
# round-6 call 9: the network's matrix waves on v_mfma_f32_16x16x32_f16 (default) against the 32x32x16 loop (variant mfma32): parity, time, counters
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call9
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_iispt_nn.py -m gpu -x -q -s > $O/tests.txt 2>&1; grep -n "passed\|failed\|within 1e-4\|Error\|error" $O/tests.txt | head -20
for rep in 1 2 3; do
for v in default mfma32; do
  if [ $v = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  timeout 600 python3 tools/net_check.py 8192 --no-torch > $O/net_check_${v}_$rep.json 2> $O/net_check_${v}_$rep.err
  python3 -c "
import json; j=json.loads(open('$O/net_check_${v}_$rep.json').readline()); print('$v', $rep, round(j['hip_net']['ms'], 3), j['n37_output_sha256'], j['fixture_err_over_max'], max(j['layers']))"
done
done
unset IILE_GPU_LIB
bash tools/net_pmc.sh default mfma32 > $O/net_pmc.log 2>&1; cp gpurun_out/net_pmc/default.txt $O/net_pmc_default.txt; cp gpurun_out/net_pmc/mfma32.txt $O/net_pmc_mfma32.txt; cat $O/net_pmc_default.txt $O/net_pmc_mfma32.txt
for v in default mfma32; do
  if [ $v = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  timeout 600 python3 bench.py --workload iispt --steps 5 --warmup 2 --cpu-seconds 0 > $O/bench_iispt_$v.json 2> $O/bench_iispt_$v.err; python3 -c "
import json; j=json.loads(open('$O/bench_iispt_$v.json').readline()); print('$v', j['ms_per_step'], j['stage_ms_per_step']['network'], j['roofline']['frac_executed'])"
done
