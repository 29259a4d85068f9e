"""SURVEY.md §8 f4, the parity half: the oracle's restatement of BVHAccel::HLBVHBuild (oracle/oracle_bvh.cpp) pinned by a
12-primitive tree worked out BY HAND from the reference's code, then the product's builders held to the oracle — the host
builder here (CPU), the device builder in tests/test_gpu_bvh_build.py.

The hand-derived case (maxPrimsInNode = 2). Twelve cubes of half-width h = 2^-8 whose centroids c are

    number  centroid                 number  centroid                number  centroid
       0    (1, 1, 1)                   4    C = (33/1024, 0, 0)        8    E = (0, 0, 1/32)
       1    A = (0, 0, 0)               5    (0, 1, 0)                  9    (0, 0, 1)
       2    B = (1/32, 0, 0)            6    D = (0, 1/32, 0)          10    (1, 0, 1)
       3    (1, 0, 0)                   7    (1, 1, 0)                 11    (0, 1, 1)

* BVHPrimitiveInfo (bvh.cpp:50-59): centroid = .5f * pMin + .5f * pMax = c exactly (dyadic numbers).
* HLBVHBuild (bvh.cpp:410-424): the centroid bounds are [0, 1]^3, so Offset() is c itself and c * 1024 is one of 0, 32, 33,
  1024 per axis. LeftShift3 (bvh.cpp:107-130): 1024 -> 1023 -> 0x09249249 (ten bits, every third), 32 -> 0x8000, 33 -> 0x8001;
  EncodeMorton3 = LeftShift3(z) << 2 | LeftShift3(y) << 1 | LeftShift3(x):
      A 0   B 0x8000   C 0x8001   D 0x10000   E 0x20000   (1,0,0) 0x09249249   (0,1,0) 0x12492492   (1,1,0) 0x1B6DB6DB
      (0,0,1) 0x24924924   (1,0,1) 0x2DB6DB6D   (0,1,1) 0x36DB6DB6   (1,1,1) 0x3FFFFFFF
* RadixSort (bvh.cpp:140-181) orders them as listed: primitive numbers 1, 2, 4, 6, 8, 3, 5, 7, 9, 10, 11, 0.
* Treelets (bvh.cpp:430-446, mask 0x3ffc0000 = Morton bits 29..18): A..E share zero top bits -> ONE treelet of five; the seven
  other corners differ there pairwise -> seven treelets of one primitive, each a leaf at once (emitLBVH: 1 < maxPrimsInNode).
* emitLBVH on A..E from bit 17 (bvh.cpp:474-553), nodes numbered in emission order:
      bit 17 (axis 17 % 3 = 2): only E has it; the binary search ends at splitOffset 4           -> T0 = interior, axis 2
        left A..D, bit 16 (axis 1): only D has it, splitOffset 3                                   -> T1 = interior, axis 1
          left A..C, bit 15 (axis 0): B and C have it, splitOffset 1                               -> T2 = interior, axis 0
            left {A}: 1 < 2                                                                        -> T3 = leaf, first 0
            right {B, C}: 2 is not < 2; bits 14..1 are equal in both (the recursion just descends); bit 0 (axis 0) differs,
              splitOffset 1                                                                        -> T4 = interior, axis 0
              {B}                                                                                  -> T5 = leaf, first 1
              {C} (bitIndex -1)                                                                    -> T6 = leaf, first 2
          right {D}                                                                                -> T7 = leaf, first 3
        right {E}                                                                                  -> T8 = leaf, first 4
  then the seven single leaves take firstPrimOffset 5..11 in treelet order (one thread).
* buildUpperSAH (bvh.cpp:555-638) over R0 = T0 (box [-h, 33/1024 + h] x [-h, 1/32 + h]^2, centroid (33/2048, 1/64, 1/64)) and
  the corner leaves R1..R7 = (1,0,0), (0,1,0), (1,1,0), (0,0,1), (1,0,1), (0,1,1), (1,1,1):
      all eight: centroid bounds [0,1]^3, extents equal -> MaximumExtent() = 2 (geometry.h:787-795: x only if strictly largest,
        then y only if > z); z-buckets: R0..R3 -> 0 (12 * 1/64 = 0.1875 -> 0), R4..R7 -> 12 -> 11; the eleven costs are equal, the
        first wins (strict <): split after bucket 0                                                -> U0 = interior, axis 2
        {R0..R3}: extents (1, 1, 1/64) -> y; R0, R1 -> bucket 0, R2, R3 -> 11                       -> U1 = interior, axis 1
          {R0, R1}: extents (1 - 33/2048, 1/64, 1/64) -> x                                          -> U2 = interior, axis 0: R0 | R1
          {R2, R3}: extents (1, 0, 0) -> x                                                          -> U3 = interior, axis 0: R2 | R3
        {R4..R7}: extents (1, 1, 0) -> y                                                            -> U4 = interior, axis 1
          {R4, R5} -> x                                                                             -> U5 = interior, axis 0
          {R6, R7} -> x                                                                             -> U6 = interior, axis 0
  (std::partition finds every range already partitioned: nothing moves.)
* flattenBVHTree (bvh.cpp:640-658), depth first, second child = `offset`:
       0 U0 ->16 |  1 U1 ->13 |  2 U2 ->12 |  3 T0 ->11 |  4 T1 ->10 |  5 T2 ->7 |  6 leaf A |  7 T4 ->9 |  8 leaf B |  9 leaf C |
      10 leaf D  | 11 leaf E  | 12 leaf R1 | 13 U3 ->15 | 14 leaf R2 | 15 leaf R3 | 16 U4 ->20 | 17 U5 ->19 | 18 leaf R4 | 19 leaf R5 |
      20 U6 ->22 | 21 leaf R6 | 22 leaf R7                                       23 nodes = 12 leaves + 11 interior
"""
import numpy as np

H = 2.0 ** -8
CENTROIDS = np.array([(1, 1, 1), (0, 0, 0), (1 / 32, 0, 0), (1, 0, 0), (33 / 1024, 0, 0), (0, 1, 0), (0, 1 / 32, 0), (1, 1, 0),
                      (0, 0, 1 / 32), (0, 0, 1), (1, 0, 1), (0, 1, 1)], np.float32)
EXPECTED_CODES = [0, 0x8000, 0x8001, 0x10000, 0x20000, 0x09249249, 0x12492492, 0x1B6DB6DB, 0x24924924, 0x2DB6DB6D, 0x36DB6DB6, 0x3FFFFFFF]
EXPECTED_ORDER = [1, 2, 4, 6, 8, 3, 5, 7, 9, 10, 11, 0]
# (is_leaf, axis or first primitive, second child or primitive count), in flattened order
EXPECTED_NODES = [(0, 2, 16), (0, 1, 13), (0, 0, 12), (0, 2, 11), (0, 1, 10), (0, 0, 7), (1, 0, 1), (0, 0, 9), (1, 1, 1), (1, 2, 1),
                  (1, 3, 1), (1, 4, 1), (1, 5, 1), (0, 0, 15), (1, 6, 1), (1, 7, 1), (0, 1, 20), (0, 0, 19), (1, 8, 1), (1, 9, 1),
                  (0, 0, 22), (1, 10, 1), (1, 11, 1)]


def _hand_case_bounds():
    return np.concatenate([CENTROIDS - np.float32(H), CENTROIDS + np.float32(H)], axis=1).astype(np.float32)


def _expected_boxes(b6):
    """Boxes of the hand-derived tree, bottom-up: a leaf's is the union of its primitives' (bvh.cpp:486-492), an interior node's
    the union of its children's (InitInterior, bvh.cpp:72-79)."""
    sorted_b = b6[EXPECTED_ORDER]
    n = len(EXPECTED_NODES)
    bmin, bmax = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    for i in range(n - 1, -1, -1):
        leaf, a, b = EXPECTED_NODES[i]
        if leaf:
            s = sorted_b[a:a + b]
            bmin[i], bmax[i] = s[:, :3].min(axis=0), s[:, 3:].max(axis=0)
        else:
            bmin[i], bmax[i] = np.minimum(bmin[i + 1], bmin[b]), np.maximum(bmax[i + 1], bmax[b])
    return bmin, bmax


def check_hand_case(nodes, order, codes=None):
    if codes is not None:
        assert [int(c) for c in codes] == EXPECTED_CODES
    assert order.tolist() == EXPECTED_ORDER
    assert len(nodes) == len(EXPECTED_NODES)
    for i, (leaf, a, b) in enumerate(EXPECTED_NODES):
        nd = nodes[i]
        if leaf:
            assert (int(nd["nprims"]), int(nd["offset"])) == (b, a), f"node {i}"
        else:
            assert (int(nd["nprims"]), int(nd["axis"]), int(nd["offset"])) == (0, a, b), f"node {i}"
    bmin, bmax = _expected_boxes(_hand_case_bounds())
    assert np.array_equal(nodes["bmin"], bmin) and np.array_equal(nodes["bmax"], bmax)


def same_tree(a, b):
    assert len(a) == len(b)
    for f in ("offset", "nprims", "axis"):
        assert np.array_equal(a[f], b[f]), f
    assert np.array_equal(a["bmin"], b["bmin"]) and np.array_equal(a["bmax"], b["bmax"])


def prim_bounds(tri_p):
    p = tri_p.reshape(-1, 3, 3)
    return np.concatenate([p.min(axis=1), p.max(axis=1)], axis=1).astype(np.float32)


def test_oracle_hlbvh_matches_the_hand_derived_tree(binding, oracle):
    nodes, order, codes = oracle.bvh_hlbvh(_hand_case_bounds(), 2)
    check_hand_case(nodes, order, codes)


def test_oracle_hlbvh_edge_cases(binding, oracle):
    # no primitives: BVHAccel returns before building (bvh.cpp:189)
    nodes, order, _ = oracle.bvh_hlbvh(np.zeros((0, 6), np.float32), 4)
    assert len(nodes) == 0
    # one primitive: one treelet, one leaf, no upper tree
    nodes, order, _ = oracle.bvh_hlbvh(np.array([[0, 0, 0, 1, 1, 1]], np.float32), 4)
    assert len(nodes) == 1 and nodes[0]["nprims"] == 1 and order.tolist() == [0]
    # equal centroids: Offset() divides by nothing (pMax > pMin fails), every code is 0, emitLBVH runs out of bits (bitIndex -1)
    # and makes ONE leaf of all 37 in input order (the sort is stable)
    same = np.tile(np.array([[0, 0, 0, 2, 2, 2]], np.float32), (37, 1))
    nodes, order, codes = oracle.bvh_hlbvh(same, 4)
    assert len(nodes) == 1 and nodes[0]["nprims"] == 37 and order.tolist() == list(range(37)) and not codes.any()
    # maxPrimsInNode above 255 is clamped (bvh.cpp:185): a treelet of 300 primitives with distinct codes still splits once
    rng = np.random.default_rng(3)
    c = (rng.random((300, 3)) * (1 / 17)).astype(np.float32)  # all inside one treelet cell (1/16 of the extent per axis)
    far = np.array([[1, 1, 1]], np.float32)
    cc = np.concatenate([c, far])
    b6 = np.concatenate([cc - 0.001, cc + 0.001], axis=1).astype(np.float32)
    n1, _, codes = oracle.bvh_hlbvh(b6, 255)
    n2, _, _ = oracle.bvh_hlbvh(b6, 10 ** 6)
    same_tree(n1, n2)
    assert len(np.unique(codes >> 18)) == 2 and int(n1["nprims"].max()) <= 255 and int((n1["nprims"] > 0).sum()) >= 3


def oracle_build_hook(oracle, captured=None):
    """The oracle's builder with the signature of iile_host_overrides::bvh_build (include/iile_host.h), as a ctypes callback:
    the host loader then flattens the scene around the ORACLE's tree (and `captured` receives the primitives' world bounds in
    the loader's order — spheres included, which scene.bvh() does not show)."""
    import ctypes
    f = oracle.lib.oracle_bvh_hlbvh
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p, ctypes.c_void_p]
    proto = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32),
                             ctypes.c_void_p, ctypes.c_void_p)

    def hook(n, b6, max_prims, nodes, n_nodes, order, user):
        if captured is not None:
            captured.append((np.ctypeslib.as_array(ctypes.cast(b6, ctypes.POINTER(ctypes.c_float)), shape=(n, 6)).copy(), int(max_prims)))
        return f(n, b6, max_prims, nodes, n_nodes, order, None)

    return proto(hook)


def scenes_equal(a, b):
    an, at, ashape = a.bvh()
    bn, bt, bshape = b.bvh()
    same_tree(an, bn)
    assert np.array_equal(at.view(np.uint32), bt.view(np.uint32)) and np.array_equal(ashape, bshape)


def test_host_builder_matches_the_oracle(binding, oracle, tmp_path):
    """The product's host HLBVH builder (csrc/host/bvh_build.cpp, split method "hlbvh") against the oracle, on what the loader
    really hands a builder: killeroo-simple (66 532 triangles + the light's sphere) and a room whose walls share Morton cells.
    The oracle is plugged in through the loader's own build hook, so both scenes are flattened by the same code around the
    two trees: nodes, primitive order and shape table must agree."""
    import boxroom
    path = tmp_path / "room.pbrt"
    path.write_text(boxroom.boxroom_pbrt(ico_levels=3, n_blobs=6, wall_n=16, xres=32, yres=32, spp=1))
    for kw in (dict(xres=32, yres=32, spp=1), dict(path=str(path))):
        seen = []
        hook = oracle_build_hook(oracle, seen)
        by_oracle = binding.HostScene(accel_split="hlbvh", bvh_on_device=hook, **kw)
        by_host = binding.HostScene(accel_split="hlbvh", **kw)
        assert len(seen) == 1 and seen[0][1] == 4 and len(seen[0][0]) == by_host.info["n_primitives" if "n_primitives" in by_host.info else "n_triangles"] + (
            0 if "n_primitives" in by_host.info else by_host.info.get("n_spheres", 0))
        scenes_equal(by_oracle, by_host)
