"""Wall time of small frames (launch-bound regime): BASELINE config 0's 400x400 x 8 spp and a 128x96 x 4 spp frame."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge
b = ge._load_binding()
for (x, y, spp) in ((400, 400, 8), (128, 96, 4), (1920, 1080, 1)):
    scene = b.HostScene(xres=x, yres=y, spp=spp)
    gpu = b.GpuScene(scene)
    gpu.render()
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        film, st = gpu.render()
        ts.append(time.perf_counter() - t)
    print(f"{x}x{y}x{spp}: wall {min(ts)*1e3:.2f} ms, device {st['ms_total']:.2f} ms, passes {st['n_passes']}")
