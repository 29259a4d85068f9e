#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ into the tracked summaries under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
   profiles/<tag>_pmc_traffic.json   per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE reports half the bytes of
16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section), so reads are doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(repo, "gpurun_out", f"prof_{tag}")
dst = os.path.join(repo, "profiles")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
b = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(b) and os.path.getsize(b):
    shutil.copy(b, os.path.join(dst, f"{tag}_bench_under_profiler.json"))


def short(name):
    return name.split("(")[0].replace("void iile::", "").replace("iile::", "")


out = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0})
for kind in ("fetch", "write"):
    for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            c = r["Counter_Name"]
            out[k][c + "_KiB"] += float(r["Counter_Value"])
            if kind == "fetch" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                out[k]["launches"] += 1
res = {}
for k, v in out.items():
    n = max(v["launches"], 1)
    rd = 2 * v["FETCH_SIZE_KiB"] * 1024 / n
    wr = v["WRITE_SIZE_KiB"] * 1024 / n
    res[k] = {"launches_in_step": v["launches"], "hbm_read_bytes_per_launch": int(rd),
              "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
              "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes, averaged over the launches of one step"}
json.dump({"workload": "bench.py --steps 1 --warmup 0 (killeroo-simple 1920x1080, 64 spp)", "kernels": res},
          open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
