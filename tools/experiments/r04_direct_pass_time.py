"""Seconds for 16 passes of the IISPT direct pass on killeroo-simple 1080p (film on the device), best of 3."""
import os, sys, time
import torch
torch.cuda.init()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
scene = b.HostScene(xres=1920, yres=1080, spp=1)
gpu = b.GpuScene(scene)
film = torch.zeros((1080, 1920, 4), dtype=torch.float64, device="cuda")
gpu.render_direct(1, film_device_ptr=film.data_ptr())
best = 1e9
for rep in range(3):
    film.zero_()
    torch.cuda.synchronize(); t0 = time.time()
    gpu.render_direct(16, film_device_ptr=film.data_ptr())
    torch.cuda.synchronize(); best = min(best, time.time() - t0)
print(f"{os.environ.get('IILE_GPU_LIB', 'default').split('_')[-1]}: 16 passes {best:.4f} s, film sum {float(film.sum()):.6f}")
