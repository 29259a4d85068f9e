"""Soak: the same frames many times over — every film must be the first one's, bit for bit (a race in a queue append, a stale
buffer or an ordering bug shows up as a hash that moves). usage: python tools/soak.py [path_frames=200] [iispt_frames=60]"""
import hashlib
import importlib
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402
import boxroom  # noqa: E402

b = ge._load_binding()
n_path = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_iispt = int(sys.argv[2]) if len(sys.argv) > 2 else 60
out = {}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# BASELINE config 2: killeroo-simple 1080p x 64 spp, film on the device, one hash per frame
scene = b.HostScene(xres=1920, yres=1080, spp=64)
gpu = b.GpuScene(scene)
h, w = scene.film_shape
film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
seen, t0 = {}, time.perf_counter()
for i in range(n_path):
    gpu.render(film_device_ptr=film.data_ptr(), stream=stream, want_stats=False)
    gpu.render_status(stream)
    k = sha(film.cpu().numpy())
    seen[k] = seen.get(k, 0) + 1
out["config2_frames"] = {"frames": n_path, "distinct_films": len(seen), "seconds": round(time.perf_counter() - t0, 1)}

# a mixed-material textured room, several passes per frame (the exact finish across passes), small
import tempfile
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "room.pbrt")
    open(p, "w").write(boxroom.boxroom_pbrt(xres=480, yres=270, spp=16, materials="all", light="envmap", textures=os.path.join(td, "tex"), maxdepth=7))
    room = b.HostScene(path=p)
    g2 = b.GpuScene(room)
    seen = {}
    for i in range(n_path):
        f, _ = g2.render(spp_per_pass=(0, 3, 5)[i % 3])
        k = sha(f)
        seen[k] = seen.get(k, 0) + 1
    out["textured_room_frames"] = {"frames": n_path, "distinct_films": len(seen)}

# BASELINE config 5: the IISPT frame
nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
import iispt_torch_reference as ref_mod  # noqa: E402
s1 = b.HostScene(path=os.path.join(REPO, "scenes", "killeroo-simple.pbrt"), xres=1920, yres=1080, spp=1)
g1 = b.GpuScene(s1)
torch.manual_seed(0)
pipe = nn_mod.IisptPipeline(g1, net=ref_mod.IISPTNet().eval(), binding=b)
size = 10 * frame_mod.NUMBER_TILES
n_tasks = -(-1920 // size) * -(-1080 // size)
seen, t0 = {}, time.perf_counter()
for i in range(n_iispt):
    fr = frame_mod.IisptFrame(b, g1, pipe)
    fr.run_batched(n_tasks, radius_start=10.0)
    fr.run_direct(frame_mod.DIRECT_SAMPLES)
    k = sha(fr.image().cpu().numpy())
    seen[k] = seen.get(k, 0) + 1
out["iispt_frames"] = {"frames": n_iispt, "distinct_images": len(seen), "seconds": round(time.perf_counter() - t0, 1)}
import json
print(json.dumps(out))
sys.exit(0 if all(v.get("distinct_films", v.get("distinct_images")) == 1 for v in out.values()) else 1)
