#!/bin/bash
# memory-side traffic of the HIP network's kernels (n = 8192 probes): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (they do not fit one pass on gfx950; FETCH_SIZE reports half the bytes of 16-byte-per-lane reads: doubled, as MI355X_MICROARCH.md's HBM
# section prescribes) -> gpurun_out/net_traffic/traffic.json: bytes per kernel of one 8192-probe forward, and per probe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/net_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/nt_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/nt_$c -- python3 $R/tools/net_check.py 8192 --no-torch > $O/$c.log 2>&1
  cp $(find /tmp/nt_$c -name "*counter_collection.csv" | head -1) $O/$c.csv
  cp $(find /tmp/nt_$c -name "*kernel_trace.csv" | head -1) $O/${c}_trace.csv
done
python3 - $O <<'PY'
import csv, sys, re, json, collections
O = sys.argv[1]
def load(c):
    trace = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f"{O}/{c}_trace.csv"))}
    best = {}
    for r in csv.DictReader(open(f"{O}/{c}.csv")):
        n = r["Kernel_Name"]
        if not any(k in n for k in ("k_conv3x3", "k_pool2", "k_up2", "k_net_")): continue
        m = re.search(r"(k_conv3x3<[^>]*>|k_pool2<[^>]*>|k_up2<[^>]*>|k_net_\w+)", n).group(1)
        d = trace.get(r["Dispatch_Id"], 0)
        if m not in best or d > best[m][0]: best[m] = (d, float(r["Counter_Value"]))   # the 8192-probe launch = the longest one
    return {k: v[1] for k, v in best.items()}
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {"n_probes": 8192, "unit": "bytes (FETCH_SIZE and WRITE_SIZE are reported in KiB; reads doubled)", "kernels": {}}
tot_r = tot_w = 0.0
for k in sorted(set(f) | set(w)):
    rb, wb = 2 * f.get(k, 0) * 1024, w.get(k, 0) * 1024
    mult = 1   # (encoder0.2 and decoder2.2 were one kernel until round 6: the former now carries the max-pool in its epilogue)
    tot_r += rb * mult; tot_w += wb * mult
    out["kernels"][k] = {"read": rb, "written": wb, "launches_per_forward": mult}
out["forward_read_bytes"], out["forward_written_bytes"] = tot_r, tot_w
out["per_probe_bytes"] = (tot_r + tot_w) / 8192
json.dump(out, open(f"{O}/traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("forward_read_bytes", "forward_written_bytes", "per_probe_bytes")}))
PY
