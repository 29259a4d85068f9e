#!/usr/bin/env python3
"""Generate tests/golden/c1_golden.npz from the CPU oracle.

The oracle in libm mode reproduces the reference's recorded ray / triangle-test /
hit counts exactly (tests/test_oracle_pins.py), so its per-sample values stand in
for reference outputs. The fixture holds, for killeroo-simple 400x400 x 8 spp:
  * px, py, k            256 pixels x 8 samples (seeded, incl. silhouettes and shadow edges)
  * L_libm, L_portable   per-sample radiance in both trig modes (float32 x 3)
  * nrays                per-sample {Scene::Intersect, Scene::IntersectP} call counts (portable)
  * halton               SampleDimension(index, 0..41) of the 8 samples of 6 pixels
  * film_sum, film_sha   sums and SHA-256 of the portable-mode {X,Y,Z,w} film
Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402
import oracle_binding as ob  # noqa: E402

ge.build_if_needed()
b = ge._load_binding()
scene = b.HostScene(xres=400, yres=400, spp=8)
orc = ob.Oracle()
rng = np.random.default_rng(20261002)
pix = np.stack([rng.integers(0, 400, 256), rng.integers(0, 400, 256)], 1)
px = np.repeat(pix[:, 0], 8).astype(np.int32)
py = np.repeat(pix[:, 1], 8).astype(np.int32)
k = np.tile(np.arange(8), 256).astype(np.int32)
L_libm, _ = orc.li(scene, px, py, k, trig_mode=ob.TRIG_LIBM)
L_port, nr = orc.li(scene, px, py, k, trig_mode=ob.TRIG_PORTABLE)
hp = [(0, 0), (5, 7), (127, 127), (128, 130), (399, 399), (255, 1)]
hal = np.zeros((len(hp), 8, 42), np.float32)
hidx = np.zeros((len(hp), 8), np.int64)
for i, (x, y) in enumerate(hp):
    for kk in range(8):
        hidx[i, kk] = orc.halton_index(scene, x, y, kk)
        hal[i, kk] = [orc.halton_sample(scene, hidx[i, kk], d) for d in range(42)]
film, st = orc.render(scene, trig_mode=ob.TRIG_PORTABLE)
np.savez_compressed(os.path.join(HERE, "c1_golden.npz"), px=px, py=py, k=k, L_libm=L_libm, L_portable=L_port, nrays=nr,
                    halton_pixels=np.array(hp, np.int32), halton_index=hidx, halton=hal,
                    film_sum=film.astype(np.float64).sum(axis=(0, 1)),
                    film_sha=np.frombuffer(hashlib.sha256(film.tobytes()).digest(), np.uint8),
                    counters=np.array([st["camera_rays"], st["regular_rays"], st["shadow_rays"], st["tri_tests"],
                                       st["tri_hits"], st["nodes_closest"], st["nodes_any"]], np.int64))
print("wrote c1_golden.npz", os.path.getsize(os.path.join(HERE, "c1_golden.npz")), "bytes")
