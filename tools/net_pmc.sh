#!/bin/bash
# matrix-pipe and LDS counters of the HIP network's kernels (n = 8192 probes): tools/net_pmc.sh [variant ...] -> gpurun_out/net_pmc/<variant>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/net_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  rm -rf /tmp/np_$v
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/np_$v -- python3 $R/tools/net_check.py 8192 --no-torch > $O/$v.log 2>&1
  c=$(find /tmp/np_$v -name "*counter_collection.csv" | head -1)
  t=$(find /tmp/np_$v -name "*kernel_trace.csv" | head -1)
  python3 - "$c" "$t" "$v" > $O/$v.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
trace = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(sys.argv[2]))}
# keep, per kernel name, the dispatch with the largest grid work = the 8192-probe launch (largest duration)
by = collections.defaultdict(dict)
for r in rows:
    if "k_conv3x3" not in r["Kernel_Name"]: continue
    by[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (name, did), c in by.items():
    d = trace.get(did, 0)
    if name not in best or d > best[name][0]: best[name] = (d, c)
print(f"{sys.argv[3]}: kernel | ms | clock GHz | MFMA busy frac | wave: wait_any wait_inst active | LDS conflict/active")
for name, (d, c) in sorted(best.items(), key=lambda kv: -kv[1][0]):
    m = re.search(r"k_conv3x3<([^>]*)>", name).group(1)
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / max(d, 1)          # cycles per ns
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(cyc * 1024, 1)   # per SIMD (256 CUs x 4)
    wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{m:<26} {d/1e6:7.3f} {clk:6.2f} {mf:7.3f}   {c.get('SQ_WAIT_ANY',0)/wc:5.2f} {c.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} {c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f}   {c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):5.3f}")
PY
  cat $O/$v.txt
done
