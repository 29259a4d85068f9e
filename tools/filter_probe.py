"""Times killeroo-simple 1080p x 64 spp with a wide pixel filter (sample store + gather film kernels).
usage: python tools/filter_probe.py ['PixelFilter "gaussian"']"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

line = sys.argv[1] if len(sys.argv) > 1 else 'PixelFilter "gaussian"'
b = ge._load_binding()
src = open(os.path.join(REPO, "scenes", "killeroo-simple.pbrt")).read()
assert 'Sampler "halton"' in src
path = os.path.join(REPO, "scenes", "_killeroo_filter_probe.pbrt")  # next to the geometry it Includes
open(path, "w").write(src.replace('Sampler "halton"', line + '\nSampler "halton"', 1))
try:
    scene = b.HostScene(path=path, xres=1920, yres=1080, spp=64)
    gpu = b.GpuScene(scene)
    gpu.render()
    for _ in range(2):
        t = time.time()
        film, st = gpu.render(time_kernels=True)
        print(line, "ms_total %.1f" % st["ms_total"], "film kernels %.1f ms" % st["ms_film"], "passes", st["n_passes"],
              "wall %.3f s" % (time.time() - t))
finally:
    os.remove(path)
