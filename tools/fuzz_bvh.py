"""One-off robustness sweep of the device HLBVH builder: random triangle soups (clusters, exact duplicates, flat and
degenerate triangles, very uneven densities) loaded once with the host builder and once with iile_bvh_build_hlbvh plugged
in — the flattened trees and the primitive order must be identical. usage: python tools/fuzz_bvh.py [first_seed=0] [n=30]"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
with tempfile.TemporaryDirectory() as td:
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        nt = int(rng.integers(1, 60000))
        kind = seed % 5
        c = rng.random((nt, 3)) * 10
        if kind == 1:  # a few tight clusters in a big empty box
            c = rng.random((8, 3))[rng.integers(0, 8, nt)] * 100 + rng.normal(0, 0.01, (nt, 3))
        elif kind == 2:  # many exact duplicates (equal Morton codes: leaves beyond maxnodeprims)
            c = c[rng.integers(0, max(1, nt // 50), nt)]
        elif kind == 3:  # a plane
            c[:, 2] = 1.5
        size = 10 ** rng.uniform(-4, 0, (nt, 1, 1))
        tri = c[:, None, :] + rng.normal(0, 1, (nt, 3, 3)) * size
        if kind == 4:
            tri[::7, 1] = tri[::7, 0]  # degenerate triangles
        tri = tri.astype(np.float32)
        maxp = int(rng.choice([1, 2, 4, 8, 255]))
        path = os.path.join(td, "s.pbrt")
        with open(path, "w") as f:
            f.write('LookAt 0 -30 5 5 5 5 0 0 1\nCamera "perspective"\nFilm "image" "integer xresolution" [8] "integer yresolution" [8]\n'
                    'Accelerator "bvh" "string splitmethod" ["hlbvh"] "integer maxnodeprims" [%d]\nWorldBegin\nLightSource "point"\n' % maxp)
            f.write('Shape "trianglemesh" "point P" [' + " ".join("%.9g" % v for v in tri.reshape(-1)) + '] "integer indices" [' +
                    " ".join(str(i) for i in range(3 * nt)) + ']\nWorldEnd\n')
        host = b.HostScene(path=path)
        dev = b.HostScene(path=path, bvh_on_device=True)
        hn, ht, _ = host.bvh()
        dn, dt, _ = dev.bvh()
        ok = len(hn) == len(dn) and all(np.array_equal(hn[k], dn[k]) for k in ("offset", "nprims", "axis", "bmin", "bmax")) and \
            np.array_equal(ht.view(np.uint32), dt.view(np.uint32))
        print("seed", seed, "kind", kind, nt, "triangles, maxnodeprims", maxp, len(hn), "nodes", "OK" if ok else "MISMATCH")
        bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
