#!/bin/bash
# round-6 call 4: MaxPool2d in the producing convolution's epilogue (default) against k_pool2 (variant): same bits? time? + the predicted scaling curve
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call4
mkdir -p $O
cd $R
for rep in 1 2; do
for v in default pool_kernel; do
  if [ $v = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  timeout 600 python3 tools/net_check.py 8192 --no-torch > $O/net_check_${v}_$rep.json 2> $O/net_check_${v}_$rep.err
  python3 -c "
import json; j=json.loads(open('$O/net_check_${v}_$rep.json').readline()); print('$v', $rep, j['hip_net']['ms'], j['n37_output_sha256'], j['fixture_err_over_max'], max(j['layers']))"
done
done
unset IILE_GPU_LIB
timeout 900 python3 -m pytest tests/test_iispt_nn.py -m gpu -x -q > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 900 python3 tools/predicted_scaling.py $O/predicted_scaling.json 2> $O/predicted_scaling.err
