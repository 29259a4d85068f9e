"""Where a k_extend wavefront's cycles go: the diagnostic build (-DIILE_TRAV_STAMPS: s_memtime stamps between the sections of the
persistent loop; tools/build_variant.sh tstamps "kernels_trav api" "-DIILE_TRAV_STAMPS") renders the bench frame once and prints the
per-section share of the waves' cycles, for the camera-ray build (bounce 0) and for the later bounces. Shares, not absolute times.
usage: IILE_GPU_LIB=pbrt-v3-iile_amd/lib/variants/libiile_gpu_tstamps.so python tools/trav_stamps.py [killeroo|boxroom] [shadow]
(shadow: the build made with -DIILE_SHADOW_STAMPS instead, k_shadow's sections in the first four counters)"""
import json
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
shadow = "shadow" in sys.argv[1:]
if len(sys.argv) > 1 and sys.argv[1] == "boxroom":
    import boxroom
    tmp = tempfile.NamedTemporaryFile("w", suffix=".pbrt", delete=False)
    tmp.write(boxroom.boxroom_pbrt(xres=1920, yres=1080, spp=64, ico_levels=5, n_blobs=12, wall_n=64))
    tmp.close()
    scene = b.HostScene(path=tmp.name)
else:
    scene = b.HostScene(xres=1920, yres=1080, spp=64)
gpu = b.GpuScene(scene)
gpu.render()
_, st = gpu.render(want_stats=True, time_kernels=2)
cyc = [int(x) for x in st["path_length"]]
if "iterstats" in sys.argv[1:]:
    # the build made with -DIILE_TRAV_ITERSTATS: k_extend's votes at bounces >= 1 (kernels_trav.hip, ITER_STAT)
    iv, il, ilw, lv, ll, lw, idle, refills = cyc
    votes = max(1, iv + lv)
    print(json.dumps({
        "ms_extend_one_stream": st["ms_extend"],
        "votes": {"interior": iv, "leaf": lv, "interior_share": round(iv / votes, 4)},
        "lanes_stepping_per_interior_vote": round(il / max(1, iv), 2),
        "lanes_waiting_at_a_leaf_per_interior_vote": round(ilw / max(1, iv), 2),
        "lanes_stepping_per_leaf_vote": round(ll / max(1, lv), 2),
        "lanes_waiting_at_an_interior_record_per_leaf_vote": round(lw / max(1, lv), 2),
        "idle_lanes_per_vote": round(idle / votes, 2),
        "refill_rounds": refills,
        "votes_per_refill_round": round(votes / max(1, refills), 1),
    }, indent=1))
    sys.exit(0)
names = ["refill (+ camera-ray generation at bounce 0)", "interior steps", "leaf steps", "finish + shade-queue append + loop head"]
out = {"ms_extend_one_stream": st["ms_extend"], "ms_shadow_one_stream": st["ms_shadow"]}
for label, part in ((("k_shadow, all bounces", cyc[0:4]),) if shadow else (("bounce 0 (camera-ray build)", cyc[0:4]), ("bounces >= 1", cyc[4:8]))):
    tot = sum(part) or 1
    out[label] = {"wave_cycles": tot, "sections": {n: round(c / tot, 4) for n, c in zip(names, part)}}
print(json.dumps(out, indent=1))
