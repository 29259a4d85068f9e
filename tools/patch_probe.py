import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
scene = b.HostScene(xres=1920, yres=1080, spp=64)
gpu = b.GpuScene(scene)
gpu.render()
os.environ["IILE_PATCH_DEBUG"] = "1"
for _ in range(4):
    film, st = gpu.render()
    print("ms_total", st["ms_total"])
