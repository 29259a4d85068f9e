"""How much would the traversal kernels gain from rays handed out in a coherent order?  A bound, measured with the kernel-level
probe (iile_trace_closest, uninstrumented traversal = four-wide steps): second-bounce-like rays of the closed room (origins on the
surfaces a first wave of rays hits, directions uniform on the sphere) traced (a) in random order, (b) sorted by the Morton code of
the origin, (c) by direction octant, then Morton code.  Run under `rocprofv3 --kernel-trace --stats` and read the k_trace durations in
launch order, or take the wall times printed here (they include the host <-> device copies of the probe, equal for the three).
usage: python tools/coherence_probe.py [n_rays=4000000] [killeroo|boxroom]"""
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
    scene = b.HostScene(path=tmp.name)
    centre, extent = np.array([0, 0, 0.5], np.float32), np.array([8, 8, 0.4], np.float32)
else:
    scene = b.HostScene(xres=64, yres=64, spp=1)
    centre, extent = np.array([0, 0, 150], np.float32), np.array([300, 300, 100], np.float32)
gpu = b.GpuScene(scene)
rng = np.random.default_rng(7)


def sphere_dirs(m):
    z = rng.uniform(-1, 1, m)
    phi = rng.uniform(0, 2 * np.pi, m)
    r = np.sqrt(1 - z * z)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)


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


code = morton(o1)
octant = ((d1[:, 0] < 0).astype(np.uint64) | ((d1[:, 1] < 0).astype(np.uint64) << 1) | ((d1[:, 2] < 0).astype(np.uint64) << 2))
orders = {"random order": np.arange(m), "sorted by origin cell": np.argsort(code, kind="stable"),
          "by direction octant, then origin cell": np.argsort((octant << 32) | code, kind="stable")}
gpu.trace_closest(o1[:1000], d1[:1000], inf1[:1000], instrumented=False)
for name, idx in orders.items():
    oo, dd = np.ascontiguousarray(o1[idx]), np.ascontiguousarray(d1[idx])
    best = 1e9
    for _ in range(3):
        t = time.time()
        p, _, _ = gpu.trace_closest(oo, dd, inf1, instrumented=False)
        best = min(best, time.time() - t)
    print(f"{which}: {m} rays, {name}: {best * 1e3:.1f} ms wall (incl. copies), hits {int((p >= 0).sum())}")
