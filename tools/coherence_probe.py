"""How much would the traversal kernels gain from rays handed out in a coherent order?  A bound, measured with the kernel-level
probe (iile_trace_closest, uninstrumented traversal = four-wide steps): second-bounce-like rays of the closed room (origins on the
surfaces a first wave of rays hits, directions uniform on the sphere) traced (a) in random order, (b) sorted by the Morton code of
the origin, (c) by direction octant, then Morton code.  Run under `rocprofv3 --kernel-trace --stats` and read the k_trace durations in
launch order, or take the wall times printed here (they include the host <-> device copies of the probe, equal for the three).
usage: python tools/coherence_probe.py [n_rays=4000000] [killeroo|boxroom] [pixels]
`pixels` (round 6, VERDICT r05 "next" 4: is the pipeline's queue order worth sorting?): the first wave is the scene's own CAMERA rays in the
pipeline's path order — 16 x 16 tiles, pixel by pixel, 64 samples each, a contiguous run of tiles as the chunk cursor hands them out — so that
the second-bounce origins arrive in the order k_shade leaves them in the ray queue; traced in that order, sorted by (origin cell, direction
octant), and shuffled."""
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
which = sys.argv[2] if len(sys.argv) > 2 else "boxroom"
if which == "boxroom":
    import boxroom
    tmp = tempfile.NamedTemporaryFile("w", suffix=".pbrt", delete=False)
    tmp.write(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64))
    tmp.close()
    scene = b.HostScene(path=tmp.name, xres=1920, yres=1080, spp=64) if (len(sys.argv) > 3 and sys.argv[3] == "pixels") else b.HostScene(path=tmp.name)
    centre, extent = np.array([0, 0, 0.5], np.float32), np.array([8, 8, 0.4], np.float32)
else:
    scene = b.HostScene(xres=1920, yres=1080, spp=64) if (len(sys.argv) > 3 and sys.argv[3] == "pixels") else b.HostScene(xres=64, yres=64, spp=1)
    centre, extent = np.array([0, 0, 150], np.float32), np.array([300, 300, 100], np.float32)
gpu = b.GpuScene(scene)
rng = np.random.default_rng(7)


def sphere_dirs(m):
    z = rng.uniform(-1, 1, m)
    phi = rng.uniform(0, 2 * np.pi, m)
    r = np.sqrt(1 - z * z)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)


pixel_order = len(sys.argv) > 3 and sys.argv[3] == "pixels"
if pixel_order:
    # the camera's rays in path order: tile (row-major over the 16 x 16 tiles), pixel of the tile, sample
    h, w = scene.film_shape
    spp = 64
    n_tiles = max(1, n // (256 * spp))
    tiles_x = (w + 15) // 16
    t0 = (((h + 15) // 16) // 2) * tiles_x          # start in the middle rows of the image
    tile = np.repeat(np.arange(t0, t0 + n_tiles), 256 * spp)
    pix = np.tile(np.repeat(np.arange(256), spp), n_tiles)
    px = (tile % tiles_x) * 16 + pix % 16
    py = (tile // tiles_x) * 16 + pix // 16
    keep = (px < w) & (py < h)
    pf = np.stack([px[keep] + rng.random(keep.sum()), py[keep] + rng.random(keep.sum())], 1).astype(np.float32)
    o0, d0 = gpu.camera_rays(pf)
    n = len(o0)
else:
    # first wave: from points near the middle of the scene outwards
    o0 = (centre + rng.uniform(-0.5, 0.5, (n, 3)) * extent).astype(np.float32)
    d0 = sphere_dirs(n)
inf = np.full(n, np.inf, np.float32)
prim, tb, _ = gpu.trace_closest(o0, d0, inf, instrumented=False)
hit = prim >= 0
o1 = (o0[hit] + d0[hit] * (tb[hit, 0:1] * np.float32(0.999))).astype(np.float32)
m = len(o1)
d1 = sphere_dirs(m)
inf1 = np.full(m, np.inf, np.float32)


def morton(o):
    lo, hi = o.min(0), o.max(0)
    q = np.clip(((o - lo) / (hi - lo + 1e-20) * 1023).astype(np.uint32), 0, 1023)

    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def run_orders(o1, d1, label):
    m = len(o1)
    inf1 = np.full(m, np.inf, np.float32)
    code = morton(o1)
    octant = ((d1[:, 0] < 0).astype(np.uint64) | ((d1[:, 1] < 0).astype(np.uint64) << 1) | ((d1[:, 2] < 0).astype(np.uint64) << 2))
    orders = {"random order": np.arange(m), "sorted by origin cell": np.argsort(code, kind="stable"),
              "by direction octant, then origin cell": np.argsort((octant << 32) | code, kind="stable")}
    if pixel_order:
        cell15 = code >> np.uint64(15)   # the top 15 bits of the 30-bit Morton code: a 32^3 grid over the origins' bounds
        orders = {"the pipeline's order (tile, pixel, sample)": np.arange(m),
                  "sorted by (32^3 origin cell, direction octant) — the 18-bit key of VERDICT r05 next 4": np.argsort((cell15 << np.uint64(3)) | octant, kind="stable"),
                  "sorted by the full 30-bit origin code": np.argsort(code, kind="stable"),
                  "shuffled": rng.permutation(m)}
        # what k_shade could do at no extra pass: its output rays binned by direction octant as they are appended (eight cursors per
        # wavefront), i.e. the pipeline's order with a stable sort by octant inside consecutive blocks of the queue
        for blk in (4096, 65536):
            key = (np.arange(m, dtype=np.uint64) // np.uint64(blk)) * np.uint64(8) + octant
            orders[f"the pipeline's order, octants grouped inside blocks of {blk}"] = np.argsort(key, kind="stable")
    gpu.trace_closest(o1[:1000], d1[:1000], inf1[:1000], instrumented=False)
    res = None
    for name, idx in orders.items():
        oo, dd = np.ascontiguousarray(o1[idx]), np.ascontiguousarray(d1[idx])
        best = 1e9
        for _ in range(3):
            t = time.time()
            p, tb_, _ = gpu.trace_closest(oo, dd, inf1, instrumented=False)
            best = min(best, time.time() - t)
        if res is None:
            res = (p, tb_)   # (the first order is the identity: the hits in the queue's own order)
        print(f"{which} {label}: {m} rays, {name}: {best * 1e3:.1f} ms wall (incl. copies), hits {int((p >= 0).sum())}")
    return res


# bounce 1 (and, in pixel order, bounces 2 and 3: the survivors keep their place in the queue, as k_shade leaves them)
o, d = o1, d1
for depth in range(1, 4 if pixel_order else 2):
    p, tbx = run_orders(o, d, f"bounce {depth}")
    hit = p >= 0
    o = (o[hit] + d[hit] * (tbx[hit, 0:1] * np.float32(0.999))).astype(np.float32)
    d = sphere_dirs(len(o))
