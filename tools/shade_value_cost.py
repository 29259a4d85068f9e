"""What bit-exact VALUES cost k_shade (VERDICT r04 "next" 4): the library named by IILE_GPU_LIB (a build of kernels_shade.hip with
approximate division / square root / single-precision trigonometry) against the CPU oracle on the full 1080p x 64 spp frame —
per-kernel times (every kernel alone on the GPU), every counter of the instrumented step against the oracle's (did a path change?),
and the distribution of the film's relative deviation. usage: IILE_GPU_LIB=... python tools/shade_value_cost.py [spp=64]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: F401,E402  (its HIP runtime first)
import __graft_entry__ as ge  # noqa: E402
import oracle_binding as ob  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = ge._load_binding()
scene = b.HostScene(xres=1920, yres=1080, spp=spp)
gpu = b.GpuScene(scene)
gpu.render()
ms = {}
for _ in range(3):
    _, st = gpu.render(time_kernels=2)
    for k in ("ms_extend", "ms_shade", "ms_shadow", "ms_mis", "ms_total"):
        ms[k] = ms.get(k, 0.0) + st[k] / 3
film, cst = gpu.render(collect_stats=True)
plain, _ = gpu.render()
ref, ost = ob.Oracle().render(scene)
pairs = {"closest_rays": "regular_rays", "shadow_rays": "shadow_rays", "nodes_closest": "nodes_closest", "nodes_any": "nodes_any",
         "tri_tests": "tri_tests", "tri_hits": "tri_hits", "camera_rays": "camera_rays", "path_length": "path_length"}
counters_equal = {k: bool(cst[k] == ost[v]) for k, v in pairs.items()}
rgb, rgb_ref = scene.film_to_rgb(plain).astype(np.float64), scene.film_to_rgb(ref).astype(np.float64)
rel = np.abs(rgb - rgb_ref) / np.maximum(np.abs(rgb_ref), 1e-3)   # relative to the pixel's value (floor: 1e-3 of unit radiance)
pix = rel.max(axis=2)
print(json.dumps({"lib": os.environ.get("IILE_GPU_LIB", "in-tree build"), "spp": spp, "ms_one_stream": {k: round(v, 3) for k, v in ms.items()},
                  "counters_equal_to_the_oracle": counters_equal, "all_counters_equal": all(counters_equal.values()),
                  "instrumented_film_bitwise": bool(np.array_equal(film.view(np.uint32), ref.view(np.uint32))),
                  "plain_film_bitwise": bool(np.array_equal(plain.view(np.uint32), ref.view(np.uint32))),
                  "pixels_within_1e-4_relative": float((pix <= 1e-4).mean()), "pixels_within_1e-3": float((pix <= 1e-3).mean()),
                  "relative_deviation_percentiles": {p: float(np.percentile(pix, p)) for p in (50, 90, 99, 99.9, 100)},
                  "image_mean": float(rgb.mean()), "image_mean_oracle": float(rgb_ref.mean())}))
