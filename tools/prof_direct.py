"""The IISPT direct pass alone for profiling: python tools/prof_direct.py [PASSES=16] [REPEATS=2] [SCENE.pbrt]
(killeroo-simple at 1920 x 1080 by default). Prints the wall time of the last repeat."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
kw = {"path": sys.argv[3]} if len(sys.argv) > 3 else {}
scene = b.HostScene(xres=1920, yres=1080, spp=1, **kw)
gpu = b.GpuScene(scene)
for _ in range(rep):
    t0 = time.perf_counter()
    film = gpu.render_direct(passes)
    dt = time.perf_counter() - t0
print({"passes": passes, "ms_with_download": round(dt * 1e3, 2), "mean": float(film[..., :3].mean())})
