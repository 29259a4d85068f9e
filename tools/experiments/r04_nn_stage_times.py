"""Where the IISPT indirect pass's 'probes_and_network' stage goes: probe render, normalise, network, rescale (fp32, 1080p sweep)."""
import importlib, os, sys, time
import numpy as np, torch
torch.cuda.init()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
sys.path.insert(0, os.path.join(REPO, "tests"))
import iispt_torch_reference as ref_mod
scene = b.HostScene(xres=1920, yres=1080, spp=1)
gpu = b.GpuScene(scene)
torch.manual_seed(0)
dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
pipe = ref_mod.TorchPipeline(gpu, dtype=dtype)   # (round 4 timed the PyTorch / MIOpen network; the product path is backend="hip")
rng = np.random.default_rng(1)
n = 25058
pos = rng.uniform((-150, -100, -130), (250, 150, 0), (n, 3)).astype(np.float32)
d = rng.standard_normal((n, 3)).astype(np.float32)
H = 32
def sync():
    torch.cuda.synchronize(); return time.time()
for rep in range(3):
    t0 = sync()
    inten = torch.empty((n, H, H, 3), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(inten); dist = torch.empty((n, H, H), dtype=torch.float32, device="cuda")
    gpu.render_probes(pos, d, device_out=(inten.data_ptr(), nrm.data_ptr(), dist.data_ptr()))
    t1 = sync()
    tn = tu = tnet = 0.0
    with torch.no_grad():
        for first in range(0, n, 8192):
            sl = slice(first, min(n, first + 8192))
            a = sync(); x, means = ref_mod.normalize_downstream(inten[sl], nrm[sl], dist[sl]); bb = sync()
            y = pipe.net(x.to(dtype).contiguous(memory_format=torch.channels_last)); c = sync()
            p = ref_mod.transform_upstream(y, means); e = sync()
            tn += bb - a; tnet += c - bb; tu += e - c
    print(f"rep {rep}: render_probes {t1 - t0:.4f} s, normalise {tn:.4f}, network {tnet:.4f}, rescale {tu:.4f}")
