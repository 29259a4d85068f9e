"""Raw throughput of the IISPT probe pass (iile_render_probes, images left in HBM): N probes placed on a grid of first
hits of killeroo-simple's camera rays. usage: python tools/probe_pass_bench.py [n_probes=20736] [repeats=5]"""
import json
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20736
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
b = ge._load_binding()
scene = b.HostScene(xres=1920, yres=1080, spp=1)
gpu = b.GpuScene(scene)
# probe positions: the hemi points of the runner's schedule at radius 10 (what a frame asks for)
pos, dr = [], []
frame_mod = __import__("importlib").import_module("pbrt-v3-iile_amd.iispt_frame")
counter = 0
for (x0, y0, x1, y1, ts) in frame_mod.schedule((0, 0, 1920, 1080), 10 ** 6, 10.0):
    task = b.IisptTask(x0, y0, x1, y1, ts, counter, 0)
    v, p, d = gpu.iispt_hemi_points(task)
    sel = v.reshape(-1) == 1
    pos.append(p.reshape(-1, 3)[sel]); dr.append(d.reshape(-1, 3)[sel])
    nx, ny = task.grid()
    counter += nx * ny + (x1 - x0) * (y1 - y0)
    if sum(len(a) for a in pos) >= n or ts != 10:
        break
pos = np.concatenate(pos)[:n]; dr = np.concatenate(dr)[:n]
n = len(pos)
out = {k: torch.empty((n, 32, 32, c), dtype=torch.float32, device="cuda") for k, c in (("i", 3), ("n", 3), ("d", 1))}
dev = (out["i"].data_ptr(), out["n"].data_ptr(), out["d"].data_ptr())
gpu.render_probes(pos, dr, device_out=dev)
torch.cuda.synchronize()
best = 1e9
for _ in range(reps):
    t0 = time.time()
    _, _, _, st = gpu.render_probes(pos, dr, device_out=dev)
    torch.cuda.synchronize()
    best = min(best, time.time() - t0)
print(json.dumps({"probes": n, "best_wall_ms": round(best * 1e3, 2), "device_ms": round(st["ms_total"], 2), "probes_per_s": round(n / best),
                  "probe_pixels_per_s": round(n * 1024 / best), "passes": st["n_passes"]}))
