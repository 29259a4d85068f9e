import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
spp, per = int(sys.argv[1]), int(sys.argv[2])
scene = b.HostScene(xres=1920, yres=1080, spp=spp)
gpu = b.GpuScene(scene)
gpu.render(spp_per_pass=per)
os.environ["IILE_PATCH_DEBUG"] = "1"
t = time.time()
film, st = gpu.render(spp_per_pass=per, time_kernels=True)
print("wall", time.time() - t, "ms_total", st["ms_total"], "passes", st["n_passes"], {k: round(st[k], 1) for k in st if k.startswith("ms_")})
