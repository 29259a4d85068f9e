import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["IILE_PATCH_DEBUG"] = "1"
import __graft_entry__ as ge
b = ge._load_binding()
spp = int(sys.argv[1])
xres, yres = int(sys.argv[2]), int(sys.argv[3])
print("loading", flush=True)
scene = b.HostScene(xres=xres, yres=yres, spp=spp)
print("loaded", flush=True)
gpu = b.GpuScene(scene)
print("created", flush=True)
t = time.time()
film, st = gpu.render()
print("wall", time.time() - t, "ms_total", st["ms_total"], "passes", st["n_passes"], flush=True)
