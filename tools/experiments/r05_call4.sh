#!/bin/bash
# round-5 call 4: the whole GPU suite + smoke on the tree with the network kernels, iile_render_status, the sphere differentials
# of the direct pass and the 16-spp room test; then the headline bench line in its new form (one-stream kernel prices)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r05_call4
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_tests.txt 2>&1; tail -15 $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.txt 2>&1; tail -1 $O/bench.txt | cut -c1-1200
