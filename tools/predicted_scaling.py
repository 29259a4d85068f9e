"""The scaling curve the code PREDICTS, since none can be measured (no multi-GPU lease in six rounds; VERDICT r05 "next" 6):
every rank's shard of an N-GPU frame rendered one after the other on ONE GPU (iile_render with tile_rank / tile_nranks, film in
HBM: exactly what rank r of `bench.py --gpus N` runs before the film merge), for N = 1, 2, 4, 8 at 64 spp (config 2's frame shared
out) and 1024 spp (config 3), and the one collective priced from the link rate:

    speedup(N) = t_full / (max_r t_shard(r, N) + t_reduce(N)),   t_reduce = film bytes / 153 GB/s x (1 + (N - 2) / chunks) + launch

(ncclReduce over a ring is bound by one xGMI link: the whole film crosses each link once, pipelined in chunks.) What this cannot
see: host-side skew between ranks' launches, RCCL's real protocol choice, clocks of eight busy GPUs in one chassis.

    python tools/predicted_scaling.py [out.json]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                     # noqa: E402
import __graft_entry__ as ge     # noqa: E402

XGMI_LINK_GBS = 153.0            # per direction and link, /opt/skills/guides/MI355X_MICROARCH.md
REDUCE_LAUNCH_MS = 0.05
REDUCE_CHUNKS = 8.0


def main():
    b = ge._load_binding()
    out = {"frame": "killeroo-simple 1920x1080", "tile_map": "iile_tile_owner: (tx + ty) % N over 16x16 tiles", "cases": []}
    stream = torch.cuda.current_stream().cuda_stream
    for spp in (64, 1024):
        scene = b.HostScene(xres=1920, yres=1080, spp=spp)
        gpu = b.GpuScene(scene)
        h, w = scene.film_shape
        film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
        film_bytes = film.numel() * 4
        t_full = None
        for n in (1, 2, 4, 8):
            ms, launches = [], None
            for r in range(n):
                best = 1e30
                for _ in range(3 if spp == 64 else 2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _, st = gpu.render(tile_rank=r, tile_nranks=n, film_device_ptr=film.data_ptr(), stream=stream, want_stats=True)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) * 1e3)
                ms.append(round(best, 3))
                launches = int(st["n_passes"])
            if n == 1:
                t_full = ms[0]
            t_reduce = 0.0 if n == 1 else film_bytes / (XGMI_LINK_GBS * 1e9) * 1e3 * (1 + (n - 2) / REDUCE_CHUNKS) + REDUCE_LAUNCH_MS
            t_n = max(ms) + t_reduce
            out["cases"].append({"spp_total": spp, "n_gpus": n, "ms_per_rank_on_one_gpu": ms, "passes_per_rank": launches,
                                 "imbalance_max_over_mean": round(max(ms) / (sum(ms) / n), 4), "ms_reduce_priced": round(t_reduce, 3),
                                 "ms_frame_predicted": round(t_n, 3), "speedup_predicted": round(t_full / t_n, 3),
                                 "efficiency_predicted": round(t_full / t_n / n, 4),
                                 "work_inflation": round(sum(ms) / t_full, 4)})
        del gpu, scene, film
        torch.cuda.empty_cache()
    out["note"] = ("a PREDICTION from one GPU: the ranks' shards timed one after the other (wall clock around iile_render, best of 2-3) and the "
                   "film reduction priced at one xGMI link; `work_inflation` = sum of the shards' times over the whole frame's time: what the "
                   "fixed costs of a pass (launches, queue set-up, tails of the persistent kernels) add when the frame is cut into N")
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "predicted_scaling.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    for c in out["cases"]:
        print(json.dumps({k: c[k] for k in ("spp_total", "n_gpus", "ms_frame_predicted", "speedup_predicted", "imbalance_max_over_mean", "work_inflation")}))


if __name__ == "__main__":
    main()
