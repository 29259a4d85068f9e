"""Time what one rank of an N-GPU bench run does on a single GPU:
python tools/shard_probe.py N  -> renders tiles t % N == 0 of 1920x1080 at 64*N spp."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import __graft_entry__ as ge
b = ge._load_binding()
n = int(sys.argv[1])
scene = b.HostScene(xres=1920, yres=1080, spp=64 * n)
gpu = b.GpuScene(scene)
h, w = scene.film_shape
film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, st = gpu.render(tile_rank=0, tile_nranks=n, film_device_ptr=film.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, time_kernels=True, want_stats=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"rank 0 of {n}: {dt*1e3:.1f} ms, passes {st['n_passes']}, kernels {st['ms_total']:.1f} ms")
