#!/bin/bash
# VALU issue calibration on the GPU box (VERDICT r02 "next" 1a): tools/valu_calib.sh <tag>
#   -> gpurun_out/valu_calib_<tag>/{table.json, pmc/..counter_collection.csv, pmc_table.json}
# table.json: in-kernel cycles per instruction at 1, 2, 3, 4, 6, 8 waves per SIMD for each instruction class;
# pmc_table.json: the SQ counters of the same dispatches and the `valu_busy` figure tools/summarize_profiles.py derives
# from them, so that its saturation value can be read off for gfx950.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/valu_calib_$TAG
mkdir -p "$O"
BIN=$R/tools/_build/valu_calib
if [ ! -x "$BIN" ]; then
  mkdir -p $R/tools/_build
  hipcc -O3 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off $R/tools/valu_calib.hip -o $BIN || exit 1
fi
cd /tmp && export TMPDIR=/tmp
timeout 300 $BIN 4000 > "$O/table.json" 2> "$O/table.err"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU \
  --kernel-trace --output-format csv -d "$O/pmc" -- $BIN 1000 > "$O/pmc.log" 2>&1
python3 - <<PY
import csv, glob, json, collections, re
f = glob.glob('$O/pmc/**/*counter_collection.csv', recursive=True)
rows = collections.OrderedDict()
if f:
    for r in csv.DictReader(open(f[0])):
        m = re.search(r'k_calib<(?:\(Op\))?(\d+), (\d+)>', r['Kernel_Name'])
        if not m: continue
        key = (int(r['Dispatch_Id']), int(m.group(1)), int(m.group(2)))
        rows.setdefault(key, {})[r['Counter_Name']] = rows.get(key, {}).get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
names = json.load(open('$O/table.json'))['rows']
op_names = []
for r in names:
    if r['op'] not in op_names: op_names.append(r['op'])
out = []
seen = set()
for (did, op, k), c in rows.items():
    if (op, k) in seen: continue   # first of the three repetitions
    seen.add((op, k))
    e = {'op': op_names[op] if op < len(op_names) else op, 'waves_per_simd': k}
    e.update({n: int(v) for n, v in c.items()})
    if c.get('SQ_BUSY_CYCLES'):
        e['valu_busy_formula'] = round(4 * c.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / (c['SQ_BUSY_CYCLES'] / 32), 4)
    if c.get('SQ_WAVE_CYCLES'):
        e['wait_any_frac'] = round(c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES'], 4)
        e['wait_inst_frac'] = round(c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES'], 4)
        e['active_inst_frac'] = round(c.get('SQ_ACTIVE_INST_ANY', 0) / c['SQ_WAVE_CYCLES'], 4)
    out.append(e)
json.dump({'note': 'valu_busy_formula = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 SEs), as tools/summarize_profiles.py', 'rows': out},
          open('$O/pmc_table.json', 'w'), indent=1)
print(len(out), 'pmc rows')
PY
python3 - <<PY
import json
t = json.load(open('$O/table.json'))
print('%-44s' % 'op', ' '.join('%8s' % ('k=%d' % k) for k in (1, 2, 3, 4, 6, 8)), '  (wave-instructions per cycle per SIMD)')
ops = []
for r in t['rows']:
    if r['op'] not in ops: ops.append(r['op'])
for o in ops:
    print('%-44s' % o[:44], ' '.join('%8.3f' % r['ops_per_cyc_simd'] for r in t['rows'] if r['op'] == o))
PY
