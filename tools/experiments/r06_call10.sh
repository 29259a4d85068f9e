#!/bin/bash
# round-6 call 10: the C++ IISPT host at BASELINE's size, eight processes, every wall time quoted
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call10
mkdir -p $O
cd $R
timeout 900 python3 tools/iispt_cli_check.py $O/r06_iispt_cli_check.json > $O/cli.txt 2>&1; grep -A10 "wall_seconds" $O/cli.txt | head -14; tail -2 $O/cli.txt
