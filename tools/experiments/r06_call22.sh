#!/bin/bash
# round-6 call 22: the matrix waves fetching their weight fragments from global memory (-DNET_B_GLOBAL) against the shipped kernel (weights through LDS)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call22
mkdir -p $O
cd $R
for rep in 1 2; do
for v in default b_global; do
  if [ $v = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  timeout 600 python3 tools/net_check.py 8192 --no-torch > $O/net_check_${v}_$rep.json 2> $O/net_check_${v}_$rep.err
  python3 -c "
import json; j=json.loads(open('$O/net_check_${v}_$rep.json').readline()); print('$v', $rep, j['hip_net']['ms'], j['n37_output_sha256'], j['fixture_err_over_max'], max(j['layers']))"
done
done | tee $O/ab_b_global.txt
bash tools/net_layers.sh b_global > $O/net_layers.txt 2>&1; grep "sum of\|^ *default\|^ *b_global" $O/net_layers.txt | head -50
