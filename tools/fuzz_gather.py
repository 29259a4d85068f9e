"""One-off robustness sweep of the IISPT gather: random tasks (position, size, tile size down to 1, sampler counter, seed)
on killeroo-simple and on a mixed-material room, hemi points and gathered pixels against the oracle, bit for bit.
usage: python tools/fuzz_gather.py [first_seed=0] [n=40]"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402
import boxroom  # noqa: E402
import oracle_binding  # noqa: E402

b = ge._load_binding()
o = oracle_binding.Oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40


def same(x, y):
    x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
    return bool(((x.view(np.uint32) == y.view(np.uint32)) | (x == y)).all())


bad = 0
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "room.pbrt")
    open(path, "w").write(boxroom.boxroom_pbrt(xres=120, yres=90, spp=4, light="area", materials="mixed", textures=td))
    scenes = [b.HostScene(xres=120, yres=90, spp=4), b.HostScene(path=path)]
    gpus = [b.GpuScene(s) for s in scenes]
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        k = seed % 2
        x0, y0 = int(rng.integers(0, 100)), int(rng.integers(0, 70))
        x1, y1 = int(rng.integers(x0 + 1, 121)), int(rng.integers(y0 + 1, 91))
        ts = int(rng.integers(1, 25))
        task = b.IisptTask(x0, y0, x1, y1, ts, int(rng.integers(0, 10 ** 6)), int(rng.integers(0, 2 ** 40)))
        v, p, d = gpus[k].iispt_hemi_points(task)
        rv, rp, rd = o.iispt_hemi_points(scenes[k], task)
        ok = np.array_equal(v, rv) and same(p, rp) and same(d, rd)
        nn = rng.uniform(0.0, 2.0, v.shape + (32, 32, 3)).astype(np.float32)
        nn[rng.uniform(size=nn.shape[:4]) < 0.05] = 0
        out = gpus[k].iispt_gather(task, v, p, d, nn)
        ref = o.iispt_gather(scenes[k], task, v, p, d, nn)
        ok = ok and same(out, ref)
        print("seed", seed, ("killeroo", "room")[k], f"task ({x0},{y0})-({x1},{y1}) tile {ts}", "OK" if ok else "MISMATCH", int((out[..., 3] > 0).sum()), "pixels")
        bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
