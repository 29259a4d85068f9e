#!/bin/bash
# Is the vector memory path (TA address unit / TCP = vector L1 / TD data return) what the traversal kernels wait for?
# Two counter passes over one bench step; busy / stall cycles per kernel family against the kernel's own GRBM_GUI_ACTIVE.
#   tools/pmc_mem.sh <tag> [bench args]  -> gpurun_out/pmc_mem_<tag>/summary.json, table on stdout
set -u
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_mem_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ONE="$R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --other-steps 0 --alone-steps 0 $*"
timeout 500 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/a" -- python3 $ONE > "$O/a.log" 2>&1
timeout 500 rocprofv3 --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/b" -- python3 $ONE > "$O/b.log" 2>&1
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ('a', 'b'):
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'].split('(')[0].replace('void iile::', '').replace('iile::', '')
            if not n.startswith('k_') or '<true' in n: continue
            agg[n.split('<')[0]][r['Counter_Name'] + ('@b' if sub == 'b' and r['Counter_Name'] == 'GRBM_GUI_ACTIVE' else '')] += float(r['Counter_Value'])
out = {}
for k, v in sorted(agg.items()):
    gui = v.get('GRBM_GUI_ACTIVE', 0)
    if gui < 1e6: continue
    cyc = gui / 8.0  # summed over the 8 XCDs
    e = {n: int(x) for n, x in v.items()}
    # *_sum counters add up the 256 CUs' instances
    e['per_cu_frac_of_kernel_cycles'] = {n: round(v.get(n, 0) / 256.0 / cyc, 4) for n in ('TA_TA_BUSY_sum', 'TD_TD_BUSY_sum', 'TCP_GATE_EN1_sum', 'TCP_PENDING_STALL_CYCLES_sum')}
    cycb = v.get('GRBM_GUI_ACTIVE@b', 0) / 8.0
    if cycb:
        e['per_cu_frac_of_kernel_cycles'].update({n: round(v.get(n, 0) / 256.0 / cycb, 4) for n in ('TA_ADDR_STALLED_BY_TC_CYCLES_sum', 'TA_DATA_STALLED_BY_TC_CYCLES_sum', 'TCP_READ_TAGCONFLICT_STALL_CYCLES_sum', 'TD_TC_STALL_sum')})
    if v.get('TA_FLAT_READ_WAVEFRONTS_sum'):
        e['ta_busy_cycles_per_read_wavefront'] = round(v.get('TA_TA_BUSY_sum', 0) / v['TA_FLAT_READ_WAVEFRONTS_sum'], 1)
        e['tcp_accesses_per_read_wavefront'] = round(v.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / v['TA_FLAT_READ_WAVEFRONTS_sum'], 1)
    if v.get('TCP_TCC_READ_REQ_sum'):
        e['tcp_tcc_read_latency_cycles'] = round(v.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / v['TCP_TCC_READ_REQ_sum'], 1)
    out[k] = e
    print(k, json.dumps({x: e[x] for x in e if not x.endswith('_sum') and not x.startswith('GRBM')}))
json.dump(out, open('$O/summary.json', 'w'), indent=1)
PY
for f in "$O"/a.log "$O"/b.log; do tail -n 2 $f | cut -c1-200; done
