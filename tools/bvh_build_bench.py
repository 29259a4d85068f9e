"""f4 timing: BVHAccel's HLBVH build on the device (iile_bvh_build_hlbvh) against the host builder, per stage, and the
device-side packing of the traversal records inside iile_scene_create.
usage: python tools/bvh_build_bench.py [n_million_synthetic=4]"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
out = {}


def prim_bounds(tri_p):
    p = tri_p.reshape(-1, 3, 3)
    return np.concatenate([p.min(axis=1), p.max(axis=1)], axis=1).astype(np.float32)


def timed_load(**kw):
    t0 = time.time()
    s = b.HostScene(**kw)
    return s, time.time() - t0


for name, kw in (("killeroo-simple", dict(xres=64, yres=64, spp=1)),):
    hs, t_host = timed_load(accel_split="hlbvh", **kw)
    ds, t_dev = timed_load(accel_split="hlbvh", bvh_on_device=True, **kw)
    _, tri_p, _ = hs.bvh()
    b6 = prim_bounds(tri_p)
    b.bvh_build_hlbvh(b6, 4)
    nodes, order, st = b.bvh_build_hlbvh(b6, 4)
    t0 = time.time()
    g = b.GpuScene(hs)
    t_create = time.time() - t0
    out[name] = {"n_prims": len(b6), "scene_load_s_host_build": round(t_host, 3), "scene_load_s_device_build": round(t_dev, 3),
                 "device_build_ms": {k: round(v, 3) if isinstance(v, float) else v for k, v in st.items()},
                 "iile_scene_create_s": round(t_create, 3)}

import boxroom  # noqa: E402
import tempfile  # noqa: E402

with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "room.pbrt")
    open(path, "w").write(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64, xres=64, yres=64, spp=1))
    hs, t_host = timed_load(path=path, accel_split="hlbvh")
    ds, t_dev = timed_load(path=path, accel_split="hlbvh", bvh_on_device=True)
    _, tri_p, _ = hs.bvh()
    b6 = prim_bounds(tri_p)
    nodes, order, st = b.bvh_build_hlbvh(b6, 4)
    out["boxroom 287k"] = {"n_prims": len(b6), "scene_load_s_host_build": round(t_host, 3), "scene_load_s_device_build": round(t_dev, 3),
                           "device_build_ms": {k: round(v, 3) if isinstance(v, float) else v for k, v in st.items()}}

nm = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(nm * 1e6)
rng = np.random.default_rng(1)
# clustered soup: 64 blobs of small boxes (uneven treelets) in a unit room
centres = rng.random((64, 3)).astype(np.float32)
c = (centres[rng.integers(0, 64, n)] + rng.normal(0, 0.03, (n, 3))).astype(np.float32)
h = (rng.random((n, 3)) * 0.002 + 1e-4).astype(np.float32)
b6 = np.concatenate([c - h, c + h], axis=1).astype(np.float32)
b.bvh_build_hlbvh(b6[:1000], 4)
nodes, order, st = b.bvh_build_hlbvh(b6, 4)
out[f"synthetic soup {nm:g} M boxes"] = {"n_prims": n, "device_build_ms": {k: round(v, 3) if isinstance(v, float) else v for k, v in st.items()}}
print(json.dumps(out))
