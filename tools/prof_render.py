"""One plain render for profiling: python tools/prof_render.py XRES YRES SPP [REPEATS]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
x, y, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 1
scene = b.HostScene(xres=x, yres=y, spp=s)
gpu = b.GpuScene(scene)
for _ in range(rep):
    film, st = gpu.render()
print("ms_total", st["ms_total"])
