import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
b = ge._load_binding()
rng = np.random.default_rng(77)
nt = 800000
c = rng.random((64, 3))[rng.integers(0, 64, nt)] * 50 + rng.normal(0, 0.5, (nt, 3))
tri = (c[:, None, :] + rng.normal(0, 0.02, (nt, 3, 3))).astype(np.float32)
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "s.pbrt")
    with open(path, "w") as f:
        f.write('LookAt 0 -30 5 5 5 5 0 0 1\nCamera "perspective"\nFilm "image" "integer xresolution" [8] "integer yresolution" [8]\n'
                'Accelerator "bvh" "string splitmethod" ["hlbvh"]\nWorldBegin\nLightSource "point"\nShape "trianglemesh" "point P" [')
        np.savetxt(f, tri.reshape(-1, 9), fmt="%.9g")
        f.write('] "integer indices" [')
        np.savetxt(f, np.arange(3 * nt).reshape(-1, 3), fmt="%d")
        f.write(']\nWorldEnd\n')
    t0 = time.time(); host = b.HostScene(path=path); t1 = time.time(); dev = b.HostScene(path=path, bvh_on_device=True); t2 = time.time()
    hn, ht, _ = host.bvh(); dn, dt, _ = dev.bvh()
    ok = len(hn) == len(dn) and all(np.array_equal(hn[k], dn[k]) for k in ("offset", "nprims", "axis", "bmin", "bmax")) and np.array_equal(ht.view(np.uint32), dt.view(np.uint32))
    print(nt, "triangles", len(hn), "nodes", "IDENTICAL" if ok else "MISMATCH", "load host %.1f s, device-built %.1f s" % (t1 - t0, t2 - t1))
