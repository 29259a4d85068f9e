"""BVHAccel's HLBVH build on the device (iile_bvh_build_hlbvh) against the ORACLE's restatement of
src/accelerators/bvh.cpp:404-658 (oracle/oracle_bvh.cpp, pinned by the hand-derived tree of tests/test_oracle_bvh.py) —
the hand-derived case itself, the shipped scene and the deep-tree room through the loader's build hook, random soups through
the direct call — then against the product's host builder, and the device-side packing of the traversal records against a
numpy restatement of their layout."""
import ctypes

import numpy as np
import pytest

import test_oracle_bvh as tob

pytestmark = pytest.mark.gpu


def _same_tree(a, b):
    assert len(a) == len(b)
    for f in ("offset", "nprims", "axis"):
        assert np.array_equal(a[f], b[f]), f
    # (== on floats: a box plane may come out as -0 on one side and +0 on the other, min / max of equal values)
    assert np.array_equal(a["bmin"], b["bmin"]) and np.array_equal(a["bmax"], b["bmax"])


def _prim_bounds(tri_p):
    p = tri_p.reshape(-1, 3, 3)
    return np.concatenate([p.min(axis=1), p.max(axis=1)], axis=1).astype(np.float32)


def test_device_builder_reproduces_the_hand_derived_tree(binding):
    """The 12-primitive tree worked out from the reference's code (tests/test_oracle_bvh.py's docstring): Morton codes, the sort,
    one treelet with four interior nodes, the SAH over eight treelet roots, the flattening."""
    nodes, order, st = binding.bvh_build_hlbvh(tob._hand_case_bounds(), 2)
    tob.check_hand_case(nodes, order)
    assert st["n_treelets"] == 8 and st["n_interior"] == 11 and st["n_leaf"] == 12


def test_device_builder_matches_the_oracle_on_loaded_scenes(binding, oracle, tmp_path):
    """killeroo-simple (66 532 triangles + the light's sphere) and the 287 k-triangle room, each loaded twice through the host
    loader's build hook — once around the device builder's tree, once around the oracle's: nodes, boxes, primitive order."""
    import boxroom
    path = tmp_path / "room.pbrt"
    path.write_text(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64, xres=64, yres=64, spp=1))
    for kw in (dict(xres=64, yres=48, spp=1), dict(path=str(path))):
        by_oracle = binding.HostScene(accel_split="hlbvh", bvh_on_device=tob.oracle_build_hook(oracle), **kw)
        by_device = binding.HostScene(accel_split="hlbvh", bvh_on_device=True, **kw)
        tob.scenes_equal(by_oracle, by_device)


@pytest.mark.parametrize("seed", range(6))
def test_device_builder_matches_the_oracle_on_soups(binding, oracle, seed):
    """Direct calls on random boxes: clusters, exact duplicates (equal Morton codes: the sort must be stable), flat sheets,
    maxPrimsInNode 1 .. 255."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 20000))
    kind = seed % 3
    c = rng.random((n, 3))
    if kind == 1:
        c = c[rng.integers(0, max(n // 50, 1), n)] + rng.normal(0, 1e-3, (n, 3))  # clusters
        c[rng.integers(0, n, n // 10)] = c[0]  # exact duplicates
    if kind == 2:
        c[:, int(rng.integers(0, 3))] = 0.25  # a sheet: one axis without extent (Offset() does not divide there)
    half = rng.random((n, 3)) * 0.01
    b6 = np.concatenate([c - half, c + half], axis=1).astype(np.float32)
    max_prims = int(rng.choice([1, 2, 4, 7, 255]))
    dn, do, _ = binding.bvh_build_hlbvh(b6, max_prims)
    on, oo, _ = oracle.bvh_hlbvh(b6, max_prims)
    assert np.array_equal(do, oo)
    tob.same_tree(dn, on)


def test_upper_tree_in_short_batches_of_levels(binding, oracle, monkeypatch):
    """buildUpperSAH runs one launch per level of its recursion, a batch of levels between two looks of the host; with batches
    of two levels every scene takes the continuation path several times (depth 11-14) — the same tree."""
    rng = np.random.default_rng(77)
    n = 60000
    c = (rng.random((64, 3))[rng.integers(0, 64, n)] + rng.normal(0, 0.03, (n, 3))).astype(np.float32)
    h = (rng.random((n, 3)) * 0.002 + 1e-4).astype(np.float32)
    b6 = np.concatenate([c - h, c + h], axis=1).astype(np.float32)
    want_nodes, want_order, st = binding.bvh_build_hlbvh(b6, 4)
    on, oo, _ = oracle.bvh_hlbvh(b6, 4)
    assert np.array_equal(want_order, oo)
    tob.same_tree(want_nodes, on)
    assert st["n_treelets"] > 64
    for batch in ("1", "2", "5"):
        monkeypatch.setenv("IILE_UPPER_BATCH", batch)
        nodes, order, _ = binding.bvh_build_hlbvh(b6, 4)
        assert np.array_equal(order, want_order)
        _same_tree(nodes, want_nodes)


def test_killeroo_tree_is_the_host_builders(binding):
    host = binding.HostScene(xres=64, yres=48, spp=1, accel_split="hlbvh")
    dev = binding.HostScene(xres=64, yres=48, spp=1, accel_split="hlbvh", bvh_on_device=True)
    hn, ht, hs = host.bvh()
    dn, dt, ds = dev.bvh()
    _same_tree(hn, dn)
    assert np.array_equal(ht.view(np.uint32), dt.view(np.uint32)) and np.array_equal(hs, ds)
    assert host.info["n_interior_nodes"] == dev.info["n_interior_nodes"] and host.info["n_leaf_nodes"] == dev.info["n_leaf_nodes"]


def test_device_built_tree_renders_the_same_film(binding, oracle):
    dev = binding.HostScene(xres=96, yres=64, spp=2, accel_split="hlbvh", bvh_on_device=True)
    gpu = binding.GpuScene(dev)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(dev)
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    assert st["nodes_closest"] == ost["nodes_closest"] and st["tri_tests"] == ost["tri_tests"]
    plain, _ = gpu.render()
    assert np.array_equal(plain.view(np.uint32), ref.view(np.uint32))


def test_room_tree_is_the_host_builders(binding, tmp_path):
    """The 287 k-triangle room (BASELINE config 4's stand-in): large treelets, walls whose centroids share Morton cells."""
    import boxroom
    path = tmp_path / "room.pbrt"
    path.write_text(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64, xres=64, yres=64, spp=1))
    host = binding.HostScene(path=str(path), accel_split="hlbvh")
    dev = binding.HostScene(path=str(path), accel_split="hlbvh", bvh_on_device=True)
    hn, ht, _ = host.bvh()
    dn, dt, _ = dev.bvh()
    _same_tree(hn, dn)
    assert np.array_equal(ht.view(np.uint32), dt.view(np.uint32))


@pytest.mark.parametrize("max_prims", [1, 4, 255])
def test_direct_call_on_killeroo_bounds(binding, max_prims):
    """iile_bvh_build_hlbvh called directly: the order it returns is a permutation, leaves cover [0, n) in order, every
    interior box is the union of its children's, every leaf box the union of its primitives'."""
    scene = binding.HostScene(xres=32, yres=32, spp=1)
    _, tri_p, shape = scene.bvh()
    b6 = _prim_bounds(tri_p)
    nodes, order, st = binding.bvh_build_hlbvh(b6, max_prims)
    n = len(b6)
    assert sorted(order.tolist()) == list(range(n))
    assert st["n_nodes"] == len(nodes) == st["n_interior"] + st["n_leaf"] and st["n_interior"] == st["n_leaf"] - 1
    leaves = nodes[nodes["nprims"] > 0]
    leaves = leaves[np.argsort(leaves["offset"])]  # (the upper SAH tree visits the treelets in its own order)
    assert np.array_equal(leaves["offset"], np.concatenate([[0], np.cumsum(leaves["nprims"].astype(np.int64))[:-1]]))
    assert int(leaves["nprims"].sum()) == n
    sb = b6[order]
    for i in np.random.default_rng(5).choice(len(nodes), min(2000, len(nodes)), replace=False):
        nd = nodes[i]
        if nd["nprims"] > 0:
            s = sb[nd["offset"]:nd["offset"] + nd["nprims"]]
            assert np.array_equal(nd["bmin"], s[:, :3].min(axis=0)) and np.array_equal(nd["bmax"], s[:, 3:].max(axis=0))
        else:
            a, b = nodes[i + 1], nodes[nd["offset"]]
            assert np.array_equal(nd["bmin"], np.minimum(a["bmin"], b["bmin"])) and np.array_equal(nd["bmax"], np.maximum(a["bmax"], b["bmax"]))


def test_degenerate_inputs(binding):
    # one primitive; all centroids equal (one Morton code: a single leaf, bit_index runs out); two far clusters
    one = np.array([[0, 0, 0, 1, 1, 1]], np.float32)
    nodes, order, _ = binding.bvh_build_hlbvh(one, 4)
    assert len(nodes) == 1 and nodes[0]["nprims"] == 1 and order.tolist() == [0]
    same = np.tile(np.array([[0, 0, 0, 2, 2, 2]], np.float32), (37, 1))
    nodes, order, _ = binding.bvh_build_hlbvh(same, 4)
    assert len(nodes) == 1 and nodes[0]["nprims"] == 37 and order.tolist() == list(range(37))  # (stable sort)
    rng = np.random.default_rng(2)
    c = np.concatenate([rng.random((500, 3)), rng.random((500, 3)) + 1000]).astype(np.float32)
    b6 = np.concatenate([c - 0.01, c + 0.01], axis=1).astype(np.float32)
    nodes, order, st = binding.bvh_build_hlbvh(b6, 4)
    assert st["n_treelets"] >= 2 and sorted(order.tolist()) == list(range(1000))
    nodes, order, _ = binding.bvh_build_hlbvh(np.zeros((0, 6), np.float32), 4)
    assert len(nodes) == 0
    with pytest.raises(RuntimeError, match="max_prims_in_node"):
        binding.bvh_build_hlbvh(one, 0)


def _pack_reference(nodes, ref_shift=0):
    """The two-wide and four-wide records of a flattened tree, restated with numpy."""
    interior = nodes["nprims"] == 0
    rec = np.cumsum(interior) - 1
    ref = np.where(interior, rec, ~nodes["offset"]).astype(np.int32)
    idx = np.nonzero(interior)[0]
    L, R = idx + 1, nodes["offset"][idx]
    wide = np.zeros((len(idx), 16), np.float32)
    wide[:, 0:3], wide[:, 3:6] = nodes["bmin"][L], nodes["bmax"][L]
    wide[:, 6:9], wide[:, 9:12] = nodes["bmin"][R], nodes["bmax"][R]
    wi = wide.view(np.int32)
    wi[:, 12], wi[:, 13], wi[:, 14] = ref[L], ref[R], nodes["axis"][idx]
    w4 = np.zeros((len(idx), 32), np.float32)
    w4i = w4.view(np.int32)
    w4[:, 0:12] = np.inf
    w4[:, 12:24] = -np.inf
    meta = nodes["axis"][idx].astype(np.uint32) & 3
    for side, C in ((0, L), (1, R)):
        leaf = nodes["nprims"][C] > 0
        g = [np.where(leaf, C, C + 1), np.where(leaf, -1, nodes["offset"][C])]
        for j in range(2):
            ok = g[j] >= 0
            slot = 2 * side + j
            gi = np.where(ok, g[j], 0)
            for c in range(3):
                w4[ok, 4 * c + slot] = nodes["bmin"][gi, c][ok]
                w4[ok, 4 * (3 + c) + slot] = nodes["bmax"][gi, c][ok]
            w4i[ok, 24 + slot] = ref[gi][ok]
        meta |= np.where(leaf, 0, (nodes["axis"][C].astype(np.uint32) & 3) << (2 + 2 * side)).astype(np.uint32)
    w4i[:, 28] = meta.astype(np.int32)
    if ref_shift:  # the refs carry the axes in their low two bits: slot 0 that of the node, slots 1 and 2 those of its children
        axes = np.stack([meta & 3, (meta >> 2) & 3, (meta >> 4) & 3, np.zeros_like(meta)], axis=1).astype(np.int64)
        w4i[:, 24:28] = ((w4i[:, 24:28].astype(np.int64) << ref_shift) | axes).astype(np.int32)
    return wide, w4


@pytest.mark.parametrize("split", ["sah", "hlbvh"])
def test_wide_records_packed_on_the_device(binding, split):
    scene = binding.HostScene(xres=32, yres=32, spp=1, accel_split=split)
    nodes, _, _ = scene.bvh()
    wide, wide4, nested = binding.bvh_pack_probe(nodes)
    lib = binding.gpu_lib()
    lib.iile_wide_ref_shift.restype = ctypes.c_int32
    rw, rw4 = _pack_reference(nodes, int(lib.iile_wide_ref_shift()))
    assert nested
    assert np.array_equal(wide.view(np.uint32), rw.view(np.uint32))
    assert np.array_equal(wide4.view(np.uint32), rw4.view(np.uint32))
    # a child box that sticks out of its parent's is reported (the four-wide step is then not used)
    bad = nodes.copy()
    k = int(np.nonzero(bad["nprims"] == 0)[0][3])
    bad["bmax"][k + 1][0] = bad["bmax"][k][0] + 1
    assert not binding.bvh_pack_probe(bad)[2]
    # a tree whose links leave the array, or whose interior count is not the buffers' size, is refused (not packed blindly)
    broken = nodes.copy()
    broken["offset"][k] = len(nodes) + 5
    with pytest.raises(RuntimeError, match="child index"):
        binding.bvh_pack_probe(broken)


def test_cpp_host_with_the_device_builder(binding, tmp_path):
    """iile_pbrt --splitmethod hlbvh --bvh-device (the C++ host plugging iile_bvh_build_hlbvh into the loader) writes the
    image the Python binding renders from the host-built HLBVH tree."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "pbrt-v3-iile_amd", "lib", "iile_pbrt")
    out = tmp_path / "cli.pfm"
    p = subprocess.run([exe, os.path.join(repo, "scenes", "killeroo-simple.pbrt"), "--xres", "96", "--yres", "64", "--spp", "2",
                        "--splitmethod", "hlbvh", "--bvh-device", "--outfile", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    img = binding.read_image(str(out))
    scene = binding.HostScene(xres=96, yres=64, spp=2, accel_split="hlbvh")
    film, _ = binding.GpuScene(scene).render()
    rgb = scene.film_to_rgb(film)
    assert img.shape == (64, 96, 3)
    assert np.array_equal(img.view(np.uint32), rgb.view(np.uint32))
