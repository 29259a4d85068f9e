#!/bin/bash
# round-6 call 26: the GPU suite and smoke() on the final tree (call 23's copies of these two logs were overwritten by tools/evidence.sh's
# `cp profiles/<tag>_*`, which brought the committed ones of the previous tree along)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call26
mkdir -p $O
cd $R
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $O/r06c_pytest_gpu.txt 2>&1; grep -E "passed|failed" $O/r06c_pytest_gpu.txt
( time timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/r06c_smoke.txt 2>&1; head -2 $O/r06c_smoke.txt
