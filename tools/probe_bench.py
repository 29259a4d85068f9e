"""BASELINE config 5 exercised end to end on one GPU: the IISPT integrator's frame over killeroo-simple — the indirect pass
(hemi points, probe pass, in-process network, per-pixel gather), the direct pass (16 passes of DirectProgressiveIntegrator) and
the merge of the two film monitors (pbrt-v3-iile_amd/iispt_frame.py) — timed per stage.
The network has random weights (none ship with the reference), so the image is meaningless; the data flow and the cost are real.
usage: python tools/probe_bench.py [xres=1920] [yres=1080] [radius_start=10] [sweeps=1] [backend=hip|torch-f32|torch-bf16]
(hip: the product path, iile_iispt_net_*; the torch backends run the PyTorch module through MIOpen, for comparison only)
IILE_IISPT_BATCHED=0: task by task as the reference's runner; IILE_IISPT_TIMERS=1: seconds per stage (adds syncs)"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()  # before libiile_gpu touches HIP (as bench.py does)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

xres = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
yres = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
radius = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
backend = sys.argv[5] if len(sys.argv) > 5 else "hip"
dtype = torch.bfloat16 if backend == "torch-bf16" else torch.float32
b = ge._load_binding()
scene = b.HostScene(xres=xres, yres=yres, spp=1)
gpu = b.GpuScene(scene)
sys.path.insert(0, os.path.join(REPO, "tests"))
import iispt_torch_reference as ref_mod  # noqa: E402  (tests/: the PyTorch module)
nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
torch.manual_seed(0)
pipe = nn_mod.IisptPipeline(gpu, net=ref_mod.IISPTNet(), binding=b) if backend == "hip" else ref_mod.TorchPipeline(gpu, dtype=dtype)
size = int(radius) * frame_mod.NUMBER_TILES
tasks_per_sweep = -(-xres // size) * -(-yres // size)
batched = os.environ.get("IILE_IISPT_BATCHED", "1") != "0"
frame = frame_mod.IisptFrame(b, gpu, pipe)
# warm-up: MIOpen kernel selection (per batch size: tens of seconds the first time a size is seen), workspace
if batched:
    frame.run_batched(tasks_per_sweep * sweeps, radius_start=radius)
else:
    frame.run_task(0, 0, min(size, xres), min(size, yres), int(radius))
torch.cuda.synchronize()
frame = frame_mod.IisptFrame(b, gpu, pipe)
timers = {} if os.environ.get("IILE_IISPT_TIMERS") else None
t0 = time.time()
if batched:
    img = frame.run_batched(tasks_per_sweep * sweeps, radius_start=radius, timers=timers)
else:
    img = frame.run(tasks_per_sweep * sweeps, radius_start=radius)
torch.cuda.synchronize()
wall = time.time() - t0
frame.run_direct(1)  # (first call: allocation)
frame_d = frame_mod.IisptFrame(b, gpu, pipe)
torch.cuda.synchronize()
t0 = time.time()
frame_d.run_direct(frame_mod.DIRECT_SAMPLES)
torch.cuda.synchronize()
wall_direct = time.time() - t0
frame.film_direct.copy_(frame_d.film_direct)
t0 = time.time()
final = frame.image()
torch.cuda.synchronize()
wall_merge = time.time() - t0
st = frame.stats
rec = float((frame.film[..., 3] > 0).float().mean())
print(json.dumps({"workload": f"IISPT frame (indirect pass + 16 direct passes + merge), killeroo-simple {xres}x{yres}, radius {radius} -> tasks of {size}^2 px, {sweeps} sweep(s)",
                  "tasks": st["tasks"], "hemi_points": st["hemi_points"], "probes": st["probes"], "pixels": st["pixels"],
                  "wall_s": round(wall, 3), "probes_per_s": round(st["probes"] / wall, 1), "mpixels_gathered_per_s": round(st["pixels"] / wall / 1e6, 3),
                  "order": "task-major stages (run_batched)" if batched else "task by task (run)", "stage_seconds": timers,
                  "network_backend": backend, "pixels_with_a_sample": round(rec, 4),
                  "indirect_image_mean": float(frame.indirect_image().mean()), "finite": bool(torch.isfinite(final).all()),
                  "direct_passes": frame_mod.DIRECT_SAMPLES, "direct_wall_s": round(wall_direct, 4),
                  "direct_msamples_per_s": round(frame_mod.DIRECT_SAMPLES * xres * yres / wall_direct / 1e6, 1), "merge_wall_s": round(wall_merge, 4),
                  "frame_wall_s": round(wall + wall_direct + wall_merge, 3), "direct_image_mean": float(frame.direct_image().mean()),
                  "image_mean": float(final.mean())}))
