#!/bin/bash
# round-6 call 29: soak — the same frames many times over, every film the first one's bit for bit
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call29
mkdir -p $O
cd $R
( time timeout 1500 python3 tools/soak.py 300 80 ) > $O/soak.txt 2>&1; tail -6 $O/soak.txt
