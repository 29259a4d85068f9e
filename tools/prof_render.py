"""One plain render for profiling: python tools/prof_render.py XRES YRES SPP [REPEATS] [SCENE.pbrt]
Prints the per-kernel milliseconds of the last repeat."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
x, y, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 1
kw = {"path": sys.argv[5]} if len(sys.argv) > 5 else {}
scene = b.HostScene(xres=x, yres=y, spp=s, **kw)
gpu = b.GpuScene(scene)
for _ in range(rep):
    film, st = gpu.render(time_kernels=True)
print({k: round(v, 2) for k, v in st.items() if k.startswith("ms_")})
