#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
   profiles/<tag>_pmc_traffic.json   HBM bytes per kernel from FETCH_SIZE / WRITE_SIZE
   profiles/<tag>_pmc_lanes.json     VALU / L2 counters per kernel (SQ_*, TCC_*)
Each JSON has `kernels` (every template instantiation as the profiler names it, per launch) and `families` (the
product builds of a kernel — first template argument COUNT = false — summed over the launches of ONE bench step:
what bench.py reads). FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE reports half the bytes of
16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section), so reads are doubled; WRITE_SIZE is exact for
16-byte-per-lane stores. The PMC passes run `bench.py --steps 1 --warmup 0`: one instrumented step (COUNT builds, not
part of `families`) and exactly one timed step; kernels that have a single build for both (no template arguments) are
counted for the timed step only (the later half of their dispatches)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(repo, "gpurun_out", f"prof_{tag}")
dst = os.path.join(repo, "profiles")
os.makedirs(dst, exist_ok=True)
try:
    bench_args = open(os.path.join(src, "bench_args.txt")).read().strip()
except OSError:
    bench_args = ""
WORKLOAD = ("bench.py --steps 1 --warmup 0 --other-steps 0 " + bench_args).strip() + (
    " (synthetic 287 k-triangle closed room of tests/boxroom.py, 1920x1080, 64 spp, 1 GPU)" if "boxroom" in bench_args
    else " (killeroo-simple 1920x1080, 64 spp, 1 GPU)")
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "stats_one_stream", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_one_stream_kernel_stats.csv"))
b1 = os.path.join(src, "bench_one_stream_under_profiler.json")
if os.path.exists(b1) and os.path.getsize(b1):
    shutil.copy(b1, os.path.join(dst, f"{tag}_bench_one_stream_under_profiler.json"))
b = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(b) and os.path.getsize(b):
    shutil.copy(b, os.path.join(dst, f"{tag}_bench_under_profiler.json"))


def short(name):
    return name.split("(")[0].replace("void iile::", "").replace("iile::", "")


def family(k):
    fam = k.split("<")[0]
    args = k[k.index("<") + 1:].rstrip(">").split(",") if "<" in k else []
    product = (not args or args[0].strip() == "false") and fam != "k_generate"
    return fam, product


per_dispatch = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch id -> {counter: value}


def load(kinds):
    """{kernel: {counter: sum over dispatches, 'launches': n}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    for kind in kinds:
        for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
            seen = collections.defaultdict(set)
            rows = [r for r in csv.DictReader(open(f)) if short(r["Kernel_Name"]).startswith("k_")]
            # Kernels without template arguments (k_mis_lit, k_film_accumulate, k_film_resolve, ...) have ONE build that the
            # instrumented step launches as well: of their dispatches only the second half — the timed step's, the command
            # runs exactly one instrumented and one timed step — belongs to the step (round 3 summed both: 2x too much).
            ids = collections.defaultdict(set)
            for r in rows:
                ids[short(r["Kernel_Name"])].add(int(r["Dispatch_Id"]))
            first_kept = {k: sorted(v)[len(v) // 2] for k, v in ids.items() if "<" not in k and len(v) >= 2 and len(v) % 2 == 0}
            for r in rows:
                k = short(r["Kernel_Name"])
                if k in first_kept and int(r["Dispatch_Id"]) < first_kept[k]:
                    continue
                out[k][r["Counter_Name"]] += float(r["Counter_Value"])
                seen[k].add(r["Dispatch_Id"])
                per_dispatch[k][(kind, int(r["Dispatch_Id"]))][r["Counter_Name"]] = per_dispatch[k][(kind, int(r["Dispatch_Id"]))].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            for k, s in seen.items():
                out[k]["launches@" + kind] = len(s)
    return out


def launches_in_order(fam):
    """Counters of the product builds of `fam`, launch by launch (= bounce by bounce for the pipeline kernels): the passes
    run the same command, so the i-th dispatch of a kernel family is the same launch in every pass."""
    by_kind = collections.defaultdict(list)
    for k, d in per_dispatch.items():
        f, product = family(k)
        if f != fam or not product:
            continue
        for (kind, did), c in d.items():
            by_kind[kind].append((did, c))
    rows = []
    for kind, lst in by_kind.items():
        lst.sort()
        for i, (_, c) in enumerate(lst):
            while len(rows) <= i:
                rows.append({})
            rows[i].update({n: int(v) for n, v in c.items()})
    return rows


# ---- HBM traffic
raw = load(("fetch", "write"))
kernels, fams = {}, collections.defaultdict(lambda: {"launches_in_step": 0, "hbm_read_bytes_per_step": 0, "hbm_write_bytes_per_step": 0})
for k, v in raw.items():
    n = int(max(v.get("launches@fetch", 0), v.get("launches@write", 0), 1))
    rd = 2 * v.get("FETCH_SIZE", 0.0) * 1024
    wr = v.get("WRITE_SIZE", 0.0) * 1024
    kernels[k] = {"launches_in_step": n, "hbm_read_bytes_per_launch": int(rd / n), "hbm_write_bytes_per_launch": int(wr / n),
                  "hbm_bytes_per_launch": int((rd + wr) / n)}
    fam, product = family(k)
    if product:
        fams[fam]["launches_in_step"] += n
        fams[fam]["hbm_read_bytes_per_step"] += int(rd)
        fams[fam]["hbm_write_bytes_per_step"] += int(wr)
for f in fams.values():
    f["hbm_bytes_per_step"] = f["hbm_read_bytes_per_step"] + f["hbm_write_bytes_per_step"]
if kernels:
    json.dump({"workload": WORKLOAD,
               "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes; separate rocprofv3 --pmc passes",
               "families": fams, "kernels": kernels}, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)

# ---- VALU / L2
raw = load(("sq", "tcc", "sq2", "mem"))
kernels, fams = {}, collections.defaultdict(lambda: collections.defaultdict(float))
for k, v in raw.items():
    kernels[k] = {c: (int(x) if not c.startswith("launches@") else int(x)) for c, x in v.items()}
    fam, product = family(k)
    if product:
        for c, x in v.items():
            fams[fam][c] += x
out_f = {}
for fam, v in fams.items():
    e = {c: int(x) for c, x in v.items()}
    if v.get("SQ_ACTIVE_INST_VALU"):
        # useful lanes per issued VALU lane slot; share of the busy cycles in which a VALU instruction was in flight
        # (quad-cycle counters: x4; SQ_BUSY_CYCLES is summed over the 32 shader engines, the others over 1024 SIMDs)
        e["lane_util"] = round(v.get("SQ_THREAD_CYCLES_VALU", 0.0) / (v["SQ_ACTIVE_INST_VALU"] * 64), 4)
        if v.get("SQ_BUSY_CYCLES"):
            # NOT a utilisation out of 1: tools/valu_calib.sh (profiles/r03_valu_calib*.json) measures this expression at
            # 1.5-1.8 for a saturated VALU (0.5 wave-instructions per cycle per SIMD) and 0.76 for ONE wave per SIMD
            e["valu_busy"] = round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["SQ_BUSY_CYCLES"] / 32), 4)
            e["valu_issue_per_cycle_simd"] = round(v.get("SQ_INSTS_VALU", 0.0) / 1024 / (v["SQ_BUSY_CYCLES"] / 32), 4)  # ceiling 0.5
    if v.get("SQ_WAVE_CYCLES"):
        # where a resident wave's time goes (disjoint, MI355X_MICROARCH.md "rocprofv3 PMC slots")
        e["wave_wait_any_frac"] = round(v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4)
        e["wave_wait_inst_frac"] = round(v.get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4)
        e["wave_active_inst_frac"] = round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4)
    if v.get("GRBM_GUI_ACTIVE") and v.get("TA_TA_BUSY_sum"):
        # the vector-memory path, per CU and as a share of the kernel's own cycles (*_sum counters add up 256 CUs,
        # GRBM_GUI_ACTIVE the 8 XCDs): address unit, data return; TCP_TOTAL_CACHE_ACCESSES = L1 accesses (the lane-loads of
        # a divergent load instruction), the unit of bench.py's `vmem` roofline (tools/vmem_calib.hip)
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        e["ta_busy"] = round(v["TA_TA_BUSY_sum"] / 256.0 / cyc, 4)
        e["td_busy"] = round(v.get("TD_TD_BUSY_sum", 0.0) / 256.0 / cyc, 4)
    if v.get("TCC_HIT_sum") or v.get("TCC_MISS_sum"):
        e["l2_hit_rate"] = round(v.get("TCC_HIT_sum", 0.0) / max(v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0), 1.0), 4)
    if fam in ("k_extend", "k_shade", "k_shadow", "k_mis"):
        e["per_launch"] = launches_in_order(fam)
    out_f[fam] = e
if kernels:
    json.dump({"workload": WORKLOAD,
               "note": "counter sums over the launches of one timed bench step, product builds only (per_launch: launch by launch = bounce by bounce); lane_util = SQ_THREAD_CYCLES_VALU / "
                       "(64 x SQ_ACTIVE_INST_VALU); valu_busy = 4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 SEs)",
               "families": out_f, "kernels": kernels}, open(os.path.join(dst, f"{tag}_pmc_lanes.json"), "w"), indent=1)
print(json.dumps({"traffic_families": list(fams.keys()), "lanes": out_f}, indent=1)[:3000])
