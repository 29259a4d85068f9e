#!/bin/bash
# per-layer times of the HIP network (n = 8192 probes) for the default build and the named variants of tools/build_variant.sh:
#   tools/net_layers.sh [variant ...]   -> gpurun_out/net_layers/<variant>.csv (kernel, max ns = the 8192-probe launch)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/net_layers
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in default "$@"; do
  if [ $v = default ]; then unset IILE_GPU_LIB; else export IILE_GPU_LIB=$R/pbrt-v3-iile_amd/lib/variants/libiile_gpu_$v.so; fi
  rm -rf /tmp/nl_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nl_$v -- python3 $R/tools/net_check.py 8192 --no-torch > $O/$v.log 2>&1
  f=$(find /tmp/nl_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" "$O/$v.csv" <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"] for k in ("k_conv3x3", "k_net_", "k_pool2", "k_up2"))]
tot = 0.0
out = open(sys.argv[3], "w")
for r in rows:
    m = re.search(r"k_conv3x3<([^>]*)>", r["Name"])
    name = m.group(1) if m else re.search(r"(k_net_\w+|k_pool2<[^>]*>|k_up2<[^>]*>)", r["Name"]).group(1)
    calls = int(r["Calls"])
    mx = float(r["MaxNs"]) / 1e6
    mult = 2 if name.startswith("32, 64, 64") else 1
    tot += mx * mult
    out.write(f"{name},{mx:.3f}\n")
    print(f"{sys.argv[2]:>10} {name:<28} {mx:8.3f} ms")
print(f"{sys.argv[2]:>10} sum of the 8192-probe launches: {tot:.2f} ms")
PY
  tail -1 $O/$v.log | cut -c1-200
done
