#!/bin/bash
# round-6 call 19: what the direct pass's kernels are made of (SQ counters): VALU issue rate, waits, lanes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_call19
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/prof_direct.py 16 2 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/prof_direct.py 16 2 > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tools/prof_direct.py 16 3 > $O/st.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % p, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f[0])):
        acc[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("SQ_INSTS_VMEM_RD", 0))):
        if "SQ_BUSY_CYCLES" in v and v["SQ_BUSY_CYCLES"] > 0:
            cyc = v["SQ_BUSY_CYCLES"] / 32
            print(k, "busy Mcycles %.1f" % (cyc / 1e6), "valu issue/cycle/simd %.3f" % (v["SQ_INSTS_VALU"] / 1024 / cyc), "lane_util %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / max(v["SQ_ACTIVE_INST_VALU"] * 64, 1)),
                  "waves/simd %.2f" % (v["SQ_WAVE_CYCLES"] / 1024 / cyc / 4 if False else v["SQ_WAVE_CYCLES"] / v["SQ_BUSY_CYCLES"] / 32), "wait_any/wave %.2f" % (v["SQ_WAIT_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1)))
        elif "SQ_INSTS_VMEM_RD" in v:
            print(k, {c: int(x) for c, x in v.items()})
PY
f=$(find $O/st -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150
