"""What the IISPT frame's sharding predicts, measured on ONE GPU (no multi-GPU node has been available): every rank's share of the
frame (iispt_frame.py: tasks by number, direct passes in blocks) rendered one after the other, timed, its monitors added up and
compared with the single-rank frame. Predicted speedup at N = t(1) / (max over ranks of t(rank) + the two monitors' all-reduce priced
at one xGMI link). usage: python tools/iispt_shard_probe.py [out.json]"""
import importlib
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402
import iispt_torch_reference as ref_mod  # noqa: E402

b = ge._load_binding()
nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
scene = b.HostScene(path=os.path.join(REPO, "scenes", "killeroo-simple.pbrt"), xres=1920, yres=1080, spp=1)
gpu = b.GpuScene(scene)
torch.manual_seed(0)
pipe = nn_mod.IisptPipeline(gpu, net=ref_mod.IISPTNet().eval(), binding=b)
size = 10 * frame_mod.NUMBER_TILES
n_tasks = -(-1920 // size) * -(-1080 // size)
XGMI_LINK_GBS = 153.0


def share(rank, nranks, reps=3):
    best, frame = 1e9, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frame = frame_mod.IisptFrame(b, gpu, pipe)
        frame.run_batched(n_tasks, radius_start=10.0, rank=rank, nranks=nranks)
        frame.run_direct(frame_mod.DIRECT_SAMPLES, rank=rank, nranks=nranks)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, frame


share(0, 1, 1)   # (allocates the workspaces)
t1, whole = share(0, 1)
out = {"frame": "IISPT frame, killeroo-simple 1920x1080, radius 10: 220 tasks, 16 direct passes", "ms_one_rank": round(t1, 2), "ranks": {}}
monitor_bytes = 2 * whole.film.numel() * 8
for n in (2, 4, 8):
    times, parts = [], []
    for r in range(n):
        t, f = share(r, n)
        times.append(t)
        parts.append(f)
    probes = [p.stats["probes"] for p in parts]
    total = parts[0].reduce_monitors(others=parts[1:])
    same = bool(torch.equal(total.film, whole.film) and torch.equal(total.film_direct, whole.film_direct))
    reduce_ms = 2 * (n - 1) / n * monitor_bytes / (XGMI_LINK_GBS * 1e9) * 1e3   # ring all-reduce over one link's bandwidth
    out["ranks"][str(n)] = {"ms_per_rank": [round(t, 2) for t in times], "probes_per_rank": probes, "monitors_equal_the_single_rank_frames": same, "all_reduce_ms_priced": round(reduce_ms, 2),
                            "predicted_speedup": round(t1 / (max(times) + reduce_ms), 2)}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
