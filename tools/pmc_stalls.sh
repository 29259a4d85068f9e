#!/bin/bash
# What the waves of the big kernels wait for: instruction-issue cycles by unit, memory / LDS / instruction-fetch latencies,
# LDS bank conflicts. Three SQ counter passes over one bench step (product builds only in the table).
#   tools/pmc_stalls.sh <tag> [bench args]   -> gpurun_out/pmc_stalls_<tag>/{p1,p2,p3}/..., table on stdout, summary.json
set -u
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_stalls_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ONE="$R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --other-steps 0 --alone-steps 0 $*"
timeout 400 rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d "$O/p1" -- python3 $ONE > "$O/p1.log" 2>&1
timeout 400 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d "$O/p2" -- python3 $ONE > "$O/p2.log" 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$O/p3" -- python3 $ONE > "$O/p3.log" 2>&1
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ('p1', 'p2', 'p3'):
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'].split('(')[0].replace('void iile::', '').replace('iile::', '')
            if not n.startswith('k_') or '<true' in n: continue
            agg[n.split('<')[0]][r['Counter_Name']] += float(r['Counter_Value'])
out = {}
for k, v in sorted(agg.items()):
    if v.get('SQ_WAVE_CYCLES', 0) < 1e8: continue
    wc = v['SQ_WAVE_CYCLES']
    e = {n: int(x) for n, x in v.items()}
    e['frac_of_wave_cycles'] = {n[3:].lower(): round(v.get(n, 0) / wc, 4) for n in ('SQ_WAIT_ANY', 'SQ_INST_CYCLES_VALU', 'SQ_INST_CYCLES_VMEM', 'SQ_INST_CYCLES_SALU', 'SQ_INST_CYCLES_SMEM', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC', 'SQ_WAIT_INST_LDS')}
    e['avg_latency_quadcycles'] = {'vmem': round(v.get('SQ_INST_LEVEL_VMEM', 0) / max(v.get('SQ_INSTS_VMEM', 1), 1), 1), 'lds': round(v.get('SQ_INST_LEVEL_LDS', 0) / max(v.get('SQ_INSTS_LDS', 1), 1), 1),
                                   'smem': round(v.get('SQ_INST_LEVEL_SMEM', 0) / max(v.get('SQ_INSTS_SMEM', 1), 1), 1), 'ifetch': round(v.get('SQ_IFETCH_LEVEL', 0) / max(v.get('SQ_IFETCH', 1), 1), 1)}
    e['lds_bank_conflict_frac'] = round(v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1), 4)
    out[k] = e
    print(k, json.dumps({x: e[x] for x in ('frac_of_wave_cycles', 'avg_latency_quadcycles', 'lds_bank_conflict_frac')}), 'ifetch', int(v.get('SQ_IFETCH', 0)), 'waves', int(v.get('SQ_WAVES', 0)))
json.dump(out, open('$O/summary.json', 'w'), indent=1)
PY
tail -2 "$O"/p*.log | cut -c1-200
