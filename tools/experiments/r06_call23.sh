#!/bin/bash
# round-6 call 23: the evidence set r06c on the final tree — GPU suite, smoke, then tools/evidence.sh (kernel traces, one-stream traces,
# PMC traffic / lanes for killeroo and the room, every bench line, the IISPT frame's trace, network counters, ten processes) and the network's traffic
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=r06c
O=$R/gpurun_out/${TAG}_evidence
mkdir -p $O
cd $R
bash tools/evidence.sh $TAG > $O/evidence.log 2>&1; tail -12 $O/evidence.log | cut -c1-400
bash tools/net_traffic.sh > $O/net_traffic.log 2>&1; cp gpurun_out/net_traffic/traffic.json $O/${TAG}_net_traffic.json; tail -1 $O/net_traffic.log
# (the two logs below are written after tools/evidence.sh: its `cp profiles/<tag>_*` brings the committed ones of the previous tree along)
( time timeout 2400 python3 -m pytest tests -m gpu -x -q ) > $O/${TAG}_pytest_gpu.txt 2>&1; tail -5 $O/${TAG}_pytest_gpu.txt | head -3
( time timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/${TAG}_smoke.txt 2>&1; head -2 $O/${TAG}_smoke.txt
ls profiles | grep "^$TAG" | head -40
mkdir -p $O/profiles_made; cp profiles/${TAG}_* $O/profiles_made/ 2>/dev/null
( time timeout 1500 python3 tools/fuzz_rooms.py 75000 840 iispt ) > $O/fuzz_rooms.txt 2>&1; tail -5 $O/fuzz_rooms.txt | head -2; grep -c "OK/OK/OK" $O/fuzz_rooms.txt; grep -c "+pb" $O/fuzz_rooms.txt; grep MISMATCH $O/fuzz_rooms.txt | head -5
