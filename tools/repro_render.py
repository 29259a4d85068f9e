import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
b = ge._load_binding()
import numpy as np
x, y, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
order = sys.argv[4] if len(sys.argv) > 4 else "sp"
scene = b.HostScene(xres=x, yres=y, spp=s)
gpu = b.GpuScene(scene)
print("created", flush=True)
for c in order:
    film, st = gpu.render(collect_stats=(c == "s"))
    print("render", c, "ok", st["closest_rays"], float(film.sum()), flush=True)
