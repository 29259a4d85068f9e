"""Experiment: the IISPT frame's direct pass on a second stream (and a second scene handle, so that the two passes do not
share a workspace) beside the indirect pass — does the traversal-bound direct pass hide under the matrix-bound network?
usage: python tools/experiments/r06_two_stream_frame.py [steps=5]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
import iispt_torch_reference as ref_mod  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scene = b.HostScene(path=os.path.join(REPO, "scenes", "killeroo-simple.pbrt"), xres=1920, yres=1080, spp=1)
gpu = b.GpuScene(scene)
gpu2 = b.GpuScene(scene)
torch.manual_seed(0)
module = ref_mod.IISPTNet().eval()
pipe = nn_mod.IisptPipeline(gpu, net=module, binding=b)
radius = 10.0
size = int(radius) * frame_mod.NUMBER_TILES
n_tasks = -(-1920 // size) * -(-1080 // size)
side = torch.cuda.Stream()


def one_stream():
    frame = frame_mod.IisptFrame(b, gpu, pipe)
    frame.run_batched(n_tasks, radius_start=radius)
    frame.run_direct(frame_mod.DIRECT_SAMPLES)
    return frame.image()


def two_streams(direct_first=True):
    frame = frame_mod.IisptFrame(b, gpu, pipe)
    main = torch.cuda.current_stream()
    side.wait_stream(main)   # the monitors were zeroed on the main stream
    if direct_first:
        gpu2.render_direct(frame_mod.DIRECT_SAMPLES, first_pass=0, film_device_ptr=frame.film_direct.data_ptr(), accumulate=False, stream=side.cuda_stream)
    frame.run_batched(n_tasks, radius_start=radius)
    if not direct_first:
        gpu2.render_direct(frame_mod.DIRECT_SAMPLES, first_pass=0, film_device_ptr=frame.film_direct.data_ptr(), accumulate=False, stream=side.cuda_stream)
    main.wait_stream(side)
    return frame.image()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        img = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return img, ts


a, ta = timed(one_stream, steps)
c, tc = timed(two_streams, steps)
a2, ta2 = timed(one_stream, steps)
print("one stream  ms/frame:", " ".join(f"{t:.1f}" for t in ta), "| again:", " ".join(f"{t:.1f}" for t in ta2))
print("two streams ms/frame:", " ".join(f"{t:.1f}" for t in tc))
print("same image bit for bit:", bool(torch.equal(a.view(torch.int32), c.view(torch.int32))))
