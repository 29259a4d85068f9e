"""Throughput of the IISPT probe pass: probes placed at the first hits of a grid of camera rays of
killeroo-simple 1920x1080 (every `step`-th pixel), each looking back along its camera ray (the reference uses the
surface normal; for timing the hemisphere's orientation does not matter).
usage: python tools/probe_bench.py [step=10]"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()  # before libiile_gpu touches HIP (as bench.py does)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

step = int(sys.argv[1]) if len(sys.argv) > 1 else 10
b = ge._load_binding()
scene = b.HostScene(xres=1920, yres=1080, spp=1)
gpu = b.GpuScene(scene)
ys, xs = np.mgrid[0:1080:step, 0:1920:step]
pf = np.stack([xs.ravel() + .5, ys.ravel() + .5], -1).astype(np.float32)
o, d = gpu.camera_rays(pf)
prim, tb, _ = gpu.trace_closest(o, d, np.full(len(o), np.inf, np.float32), instrumented=False)
hit = prim >= 0
pos = (o + d * tb[:, :1])[hit] - d[hit] * 1e-3
direction = -d[hit]
print("probes", len(pos), "of", len(o), "grid points")
gpu.render_probes(pos[:64], direction[:64])
for _ in range(3):
    t = time.time()
    inten, nrm, dist, st = gpu.render_probes(pos, direction)
    wall = time.time() - t
    print("device %.1f ms (%d passes), wall %.3f s: %.0f probes/s device, %.0f probes/s incl. host cameras + copies; "
          "%.1f M probe pixels/s" % (st["ms_total"], st["n_passes"], wall, len(pos) / st["ms_total"] * 1e3, len(pos) / wall,
                                     len(pos) * 1024 / st["ms_total"] / 1e3))
print("mean intensity %.4f, hit fraction of probe rays %.3f" % (float(inten.mean()), float((dist >= 0).mean())))
# outputs left in HBM (what an in-process network would read): torch only allocates the buffers
n = len(pos)
t_int = torch.empty((n, 32, 32, 3), dtype=torch.float32, device="cuda")
t_nrm = torch.empty((n, 32, 32, 3), dtype=torch.float32, device="cuda")
t_dst = torch.empty((n, 32, 32), dtype=torch.float32, device="cuda")
for _ in range(3):
    torch.cuda.synchronize()
    t = time.time()
    _, _, _, st = gpu.render_probes(pos, direction, device_out=(t_int.data_ptr(), t_nrm.data_ptr(), t_dst.data_ptr()))
    wall = time.time() - t
    print("outputs in HBM: device %.1f ms, wall %.3f s: %.0f probes/s" % (st["ms_total"], wall, n / wall))
assert np.array_equal(t_int.cpu().numpy(), inten) and np.array_equal(t_dst.cpu().numpy(), dist)

# the whole stage: render -> normalise -> IISPTNet (random weights: none ship with the reference) -> rescale
import importlib  # noqa: E402
nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
for dtype in (torch.float32, torch.bfloat16):
    pipe = nn_mod.IisptPipeline(gpu, dtype=dtype)
    pipe(pos[:512], direction[:512])
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.time()
        pred, _, _, _ = pipe(pos, direction, batch=4096)
        torch.cuda.synchronize()
        wall = time.time() - t
        print("render + normalise + IISPTNet (%s) + rescale: wall %.3f s, %.0f probes/s (~0.99 GFLOP per probe: %.1f TFLOP/s incl. the render)"
              % (str(dtype).split(".")[-1], wall, n / wall, n * 0.99e9 / wall / 1e12))
