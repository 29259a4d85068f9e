"""BASELINE config 1 in full: killeroo-simple 1920x1080 x 64 spp rendered by the GPU kernels bench.py times and by
the CPU oracle on all host cores, compared bit for bit (film {X, Y, Z, weight} and the traversal counters).
usage: python tools/full_frame_parity.py [out.json] [boxroom-textured SPP | boxroom SPP | killeroo SPP]   (default: killeroo-simple at 64 spp)"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
import oracle_binding  # noqa: E402

workload = "killeroo-simple 1920x1080, 64 spp, path maxdepth 5 (BASELINE.json configs[1])"
if len(sys.argv) > 2 and sys.argv[2] == "boxroom-textured":
    import tempfile
    import boxroom
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    tmp = tempfile.NamedTemporaryFile("w", suffix=".pbrt", delete=False)
    tmp.write(boxroom.boxroom_pbrt(xres=1920, yres=1080, spp=spp, ico_levels=5, n_blobs=12, wall_n=64, light="envmap", materials="mixed",
                                   textures=tempfile.mkdtemp(prefix="boxroom_img_")))
    tmp.close()
    scene = b.HostScene(path=tmp.name)
    workload = f"synthetic textured boxroom (tests/boxroom.py: environment map, image / scale textures, bump maps, alpha masks) 1920x1080, {spp} spp"
elif len(sys.argv) > 2 and sys.argv[2] == "boxroom":
    # BASELINE config 4's stand-in (Sponza does not ship with the reference): the closed 287 k-triangle room, at BASELINE's 256 spp by default
    import tempfile
    import boxroom
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    tmp = tempfile.NamedTemporaryFile("w", suffix=".pbrt", delete=False)
    tmp.write(boxroom.boxroom_pbrt(xres=1920, yres=1080, spp=spp, ico_levels=5, n_blobs=12, wall_n=64))
    tmp.close()
    scene = b.HostScene(path=tmp.name)
    workload = f"synthetic boxroom (287k triangles, tests/boxroom.py) 1920x1080, {spp} spp, path maxdepth 5 (BASELINE.json configs[3]'s stand-in)"
elif len(sys.argv) > 3 and sys.argv[2] == "killeroo":
    spp = int(sys.argv[3])
    scene = b.HostScene(xres=1920, yres=1080, spp=spp)
    workload = f"killeroo-simple 1920x1080, {spp} spp, path maxdepth 5 (BASELINE.json configs[2] is this frame at 1024 spp over 8 GPUs)"
else:
    scene = b.HostScene(xres=1920, yres=1080, spp=64)
gpu = b.GpuScene(scene)
gpu.render(k_begin=0, k_end=1)
t = time.time()
film, st = gpu.render()
t_gpu = time.time() - t
counted, cst = gpu.render(collect_stats=True)
t = time.time()
ref, ost = oracle_binding.Oracle().render(scene)
t_cpu = time.time() - t


def same(a, c):
    return bool((a.view(np.uint32) == c.view(np.uint32)).all() or ((a == c) | (np.isnan(a) & np.isnan(c))).all())


out = {
    "workload": workload,
    "camera_samples": int(ost["camera_rays"]), "rays": int(ost["regular_rays"] + ost["shadow_rays"]),
    "film_bitwise_equal_plain_kernels": same(film, ref), "film_bitwise_equal_instrumented_kernels": same(counted, ref),
    "counters_equal": all(int(cst[a]) == int(ost[o]) for a, o in (("closest_rays", "regular_rays"), ("shadow_rays", "shadow_rays"),
                                                                  ("nodes_closest", "nodes_closest"), ("nodes_any", "nodes_any"),
                                                                  ("tri_tests", "tri_tests"), ("tri_hits", "tri_hits"))) and list(cst["path_length"]) == list(ost["path_length"]),
    "gpu_wall_s": round(t_gpu, 3), "gpu_ms_total": round(st["ms_total"], 2), "oracle_wall_s": round(t_cpu, 1), "oracle_threads": int(ost["threads"]),
    "image_mean_rgb": [float(v) for v in scene.film_to_rgb(film).reshape(-1, 3).astype(np.float64).mean(0)],
}
diff = film != ref
out["differing_values"] = int(diff.sum())
if diff.any():
    idx = np.argwhere(diff)
    out["first_differences"] = [[int(i) for i in ix] + [float(film[tuple(ix)]), float(ref[tuple(ix)])] for ix in idx[:6]]
    rel = np.abs(film[diff].astype(np.float64) - ref[diff]) / np.maximum(np.abs(ref[diff]), 1e-30)
    out["max_relative_difference"] = float(rel.max())
    out["differing_pixels"] = int(diff.any(-1).sum())
print(json.dumps(out))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(json.dumps(out, indent=1) + "\n")
sys.exit(0 if out["film_bitwise_equal_plain_kernels"] and out["film_bitwise_equal_instrumented_kernels"] and out["counters_equal"] else 1)
