"""Time what EVERY rank of an N-GPU bench run does, one after the other on a single GPU:
python tools/shard_probe.py N [SPP_TOTAL]  -> renders each rank's tiles (iile_tile_owner) of 1920x1080 at SPP_TOTAL
(default 128*N, bench.py's weak mode) and prints per-rank milliseconds: the load imbalance an N-GPU run would see."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import __graft_entry__ as ge
b = ge._load_binding()
n = int(sys.argv[1])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128 * n
scene = b.HostScene(xres=1920, yres=1080, spp=spp)
gpu = b.GpuScene(scene)
h, w = scene.film_shape
film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
ms, paths = [], []
for r in range(n):
    best = 1e30
    for i in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, st = gpu.render(tile_rank=r, tile_nranks=n, film_device_ptr=film.data_ptr(), stream=stream, want_stats=True)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    ms.append(round(best, 2)); paths.append(st["n_paths"])
print(json.dumps({"n_ranks": n, "spp_total": spp, "ms_per_rank": ms, "paths_per_rank": paths,
                  "max_over_mean": round(max(ms) / (sum(ms) / n), 4), "tile_map": "iile_tile_owner: (tx + ty) % n"}))
