"""Full-size spot checks against the oracle: (1) a Gaussian pixel filter at 1920x1080 x 16 spp, film bit for bit;
(2) a batch of 20 736 probes, a sample of which is compared bit for bit with probes rendered one by one by the oracle."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
import oracle_binding  # noqa: E402

o = oracle_binding.Oracle()
src = open(os.path.join(REPO, "scenes", "killeroo-simple.pbrt")).read()
path = os.path.join(REPO, "scenes", "_killeroo_scale_check.pbrt")
open(path, "w").write(src.replace('Sampler "halton"', 'PixelFilter "gaussian"\nSampler "halton"', 1))
try:
    scene = b.HostScene(path=path, xres=1920, yres=1080, spp=16)
finally:
    os.remove(path)
film, st = b.GpuScene(scene).render()
ref, _ = o.render(scene)
same = (film.view(np.uint32) == ref.view(np.uint32)) | (film == ref)
print("gaussian 1080p x 16 spp: bitwise equal", bool(same.all()), "differing", int((~same).sum()), "passes", st["n_passes"])

scene = b.HostScene(xres=1920, yres=1080, spp=1)
gpu = b.GpuScene(scene)
ys, xs = np.mgrid[0:1080:10, 0:1920:10]
pf = np.stack([xs.ravel() + .5, ys.ravel() + .5], -1).astype(np.float32)
ro, rd = gpu.camera_rays(pf)
prim, tb, _ = gpu.trace_closest(ro, rd, np.full(len(ro), np.inf, np.float32), instrumented=False)
pos = (ro + rd * tb[:, :1]) - rd * 1e-3
direction = -rd
inten, nrm, dist, pst = gpu.render_probes(pos, direction)
ok = True
for i in (0, 1, 777, 5000, 12345, 20000, len(pos) - 1):
    oi, on, od = o.render_probe(scene, pos[i], direction[i])
    ok = ok and np.array_equal(inten[i], oi) and np.array_equal(nrm[i], on) and np.array_equal(dist[i], od)
print("probe batch of", len(pos), "probes: sampled probes bitwise equal", ok, "passes", pst["n_passes"])
