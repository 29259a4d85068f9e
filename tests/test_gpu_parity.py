"""GPU parity: every stage of the HIP path against the CPU oracle, through the C ABI.

The oracle runs in ORACLE_TRIG_PORTABLE mode, i.e. with the same double-precision
sin/cos/acos evaluation the kernels use, so all comparisons are bit-exact
(index / integer work) or bitwise on float32 (everything else). The tolerance
north_star states for radiance (1e-4 relative) is therefore met with margin;
tests that compare against the libm-mode oracle state their tolerance inline.
"""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bitwise(a, b, what):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    same = bits(a) == bits(b)
    # +0 / -0 and NaN payloads are not distinguished by the path
    same |= (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{what}: {int((~same).sum())} of {same.size} values differ, first at {np.argwhere(~same)[0]}"


def test_device_present(binding):
    assert binding.device_count() >= 1


def test_trig_matches_oracle(binding, oracle):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-7, 7, 20000), rng.uniform(-1, 1, 20000), [0.0, 1.0, -1.0, 0.5, -0.5, 6.2831855]])
    x = x.astype(np.float32)
    dev = binding.trig_probe(x)
    sc = oracle.sincos(x)
    assert_bitwise(dev[:, 0], sc[:, 0], "sin")
    assert_bitwise(dev[:, 1], sc[:, 1], "cos")
    assert_bitwise(dev[:, 2], oracle.acos(np.clip(x, -1, 1)), "acos")


def test_halton_matches_oracle(gpu_c1, scene_c1, oracle):
    # pixels / sample numbers SURVEY.md §8c lists for the Halton capture
    pix = [(0, 0), (5, 7), (127, 127), (128, 130), (399, 399), (255, 1), (17, 300)]
    ks = [0, 1, 7]
    px = np.array([p[0] for p in pix for _ in ks], np.int32)
    py = np.array([p[1] for p in pix for _ in ks], np.int32)
    k = np.array([kk for _ in pix for kk in ks], np.int32)
    ndims = 42
    dev, idx = gpu_c1.halton_samples(px, py, k, 0, ndims)
    for i in range(len(px)):
        ref_idx = oracle.halton_index(scene_c1, px[i], py[i], k[i])
        assert int(idx[i]) == ref_idx
        ref = np.array([oracle.halton_sample(scene_c1, ref_idx, d) for d in range(ndims)], np.float32)
        assert_bitwise(dev[i], ref, f"halton pixel {pix[i // len(ks)]} k {k[i]}")


def test_halton_reference_kat(gpu_c1):
    # outputs of the reference recorded in SURVEY.md §8c (sampleBounds [0,400)^2)
    dev, idx = gpu_c1.halton_samples([5, 5], [7, 7], [0, 1], 0, 10)
    assert int(idx[0]) == 20304 and int(idx[1]) == 51408
    want = np.array([0.47265625, 0.67078203, 0.130083218, 0.290784538, 0.161110461, 0.937937021, 0.528595328,
                     0.759349823, 0.738907099, 0.000506668701], np.float32)
    assert_bitwise(dev[0], want, "Halton KAT pixel (5,7) k=0")
    assert_bitwise(dev[1, :2], np.array([0.537109375, 0.539094746], np.float32), "Halton KAT k=1")


def test_camera_rays_match_oracle(gpu_c1, scene_c1, oracle):
    rng = np.random.default_rng(2)
    pf = np.stack([rng.uniform(0, 400, 1024), rng.uniform(0, 400, 1024)], 1).astype(np.float32)
    pf[:4] = [[0, 0], [400, 400], [5.4726562, 7.670782], [200, 200]]
    o, d = gpu_c1.camera_rays(pf)
    ro, rd = oracle.camera_rays(scene_c1, pf)
    assert_bitwise(o, ro, "camera origin")
    assert_bitwise(d, rd, "camera direction")


def _camera_and_random_rays(scene, oracle, n, seed):
    rng = np.random.default_rng(seed)
    h, w = scene.film_shape
    pf = np.stack([rng.uniform(0, w, n // 2), rng.uniform(0, h, n // 2)], 1).astype(np.float32)
    o1, d1 = oracle.camera_rays(scene, pf)
    # rays from random points around the scene towards the killeroos / light
    o2 = rng.uniform(-300, 300, (n - n // 2, 3)).astype(np.float32)
    tgt = rng.uniform(-150, 150, (n - n // 2, 3)).astype(np.float32)
    d2 = (tgt - o2).astype(np.float32)
    o = np.concatenate([o1, o2]).astype(np.float32)
    d = np.concatenate([d1, d2]).astype(np.float32)
    tmax = np.full(n, np.inf, np.float32)
    tmax[n // 2:] = rng.choice([np.inf, 1.0, 0.9999, 0.5], n - n // 2).astype(np.float32)
    return o, d, tmax


def _degenerate_rays(n, seed):
    """Rays whose slab products hit 0 * inf (NaN) or +-inf: directions with zero (and negative
    zero) components, origins exactly on box planes of killeroo-simple (floor z = -140, wall
    x = -400, quad extents +-1000, -1140, 860) — the cases where Bounds3::IntersectP's
    compare-and-replace chain is not a plain min/max."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-500, 500, (n, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    kind = rng.integers(0, 8, n)
    planes = {0: (2, -140.0), 1: (0, -400.0), 2: (0, 1000.0), 3: (1, -1000.0), 4: (2, 860.0), 5: (2, -1140.0)}
    for i in range(n):
        k = int(kind[i])
        if k in planes:  # on a box plane, travelling inside it
            ax, val = planes[k]
            o[i, ax] = val
            d[i, ax] = 0.0 if rng.random() < .5 else -0.0
        elif k == 6:  # one zero component, arbitrary origin
            d[i, rng.integers(0, 3)] = 0.0 if rng.random() < .5 else -0.0
        else:  # axis aligned: two zero components
            ax = rng.integers(0, 3)
            keep = d[i, ax] if d[i, ax] != 0 else 1.0
            d[i] = [0.0 if rng.random() < .5 else -0.0 for _ in range(3)]
            d[i, ax] = keep
    tmax = rng.choice([np.inf, 2000.0, 500.0], n).astype(np.float32)
    return o, d, tmax


@pytest.mark.parametrize("instrumented", [True, False])
def test_closest_hit_matches_oracle(gpu_c1, scene_c1, oracle, instrumented):
    """instrumented=False is the traversal the timed render kernels run (four-wide steps)."""
    o, d, tmax = _camera_and_random_rays(scene_c1, oracle, 65536, 3)
    o2, d2, t2 = _degenerate_rays(16384, 13)
    o, d, tmax = np.concatenate([o, o2]), np.concatenate([d, d2]), np.concatenate([tmax, t2])
    prim, tb, st = gpu_c1.trace_closest(o, d, tmax, instrumented=instrumented)
    rprim, rtb = oracle.intersect(scene_c1, o, d, tmax)
    assert np.array_equal(prim, rprim), f"{int((prim != rprim).sum())} closest-hit primitives differ"
    assert (prim >= 0).sum() > 10000 and (prim[65536:] >= 0).sum() > 1000
    assert_bitwise(tb, rtb, "closest hit (t, b0, b1, b2)")


@pytest.mark.parametrize("instrumented", [True, False])
def test_any_hit_matches_oracle(gpu_c1, scene_c1, oracle, instrumented):
    o, d, tmax = _camera_and_random_rays(scene_c1, oracle, 65536, 4)
    o2, d2, t2 = _degenerate_rays(16384, 14)
    o, d, tmax = np.concatenate([o, o2]), np.concatenate([d, d2]), np.concatenate([tmax, t2])
    hit, st = gpu_c1.trace_any(o, d, tmax, instrumented=instrumented)
    rhit = oracle.intersect_p(scene_c1, o, d, tmax)
    assert np.array_equal(hit, rhit)
    assert 1000 < hit.sum() < len(hit)


def test_bsdf_matches_oracle(gpu_c1, scene_c1, oracle):
    rng = np.random.default_rng(5)
    n = 4096

    def dirs(m):
        v = rng.normal(size=(m, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v.astype(np.float32)

    wo, wi = dirs(n), dirs(n)
    wo[:8, 2] = [0, 1e-8, -1e-8, 1, -1, 0.99995, 0.5, -0.5]
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    u[:4] = [[0, 0], [0.99999994, 0.99999994], [0.5, 0.5], [0.49999997, 0.5]]
    for mat in range(scene_c1.info["n_materials"]):
        ev = gpu_c1.bsdf_eval(mat, wo, wi)
        assert_bitwise(ev, oracle.bsdf_eval(scene_c1, mat, wo, wi), f"BSDF f/pdf material {mat}")
        sm = gpu_c1.bsdf_sample(mat, wo, u)
        assert_bitwise(sm, oracle.bsdf_sample(scene_c1, mat, wo, u), f"BSDF Sample_f material {mat}")


def test_li_per_sample_matches_oracle(gpu_c1, scene_c1, oracle):
    # all 8 samples of 256 pixels incl. silhouette / light / shadow-edge regions
    rng = np.random.default_rng(6)
    pix = np.stack([rng.integers(0, 400, 256), rng.integers(0, 400, 256)], 1)
    px = np.repeat(pix[:, 0], 8).astype(np.int32)
    py = np.repeat(pix[:, 1], 8).astype(np.int32)
    k = np.tile(np.arange(8), 256).astype(np.int32)
    L, nr = gpu_c1.li_samples(px, py, k)
    rL, rnr = oracle.li(scene_c1, px, py, k)
    assert np.array_equal(nr, rnr), "per-sample ray counts differ"
    assert_bitwise(L, rL, "per-sample radiance")
    assert (L > 0).any()


def test_film_small_bitwise_and_counters(gpu_small, scene_small, oracle):
    film, st = gpu_small.render(collect_stats=True, time_kernels=True)
    ref, ost = oracle.render(scene_small)
    assert_bitwise(film, ref, "film {X,Y,Z,w}")
    assert st["camera_rays"] == ost["camera_rays"] == 160 * 120 * 4
    assert st["closest_rays"] == ost["regular_rays"]
    assert st["shadow_rays"] == ost["shadow_rays"]
    assert st["tri_tests"] == ost["tri_tests"] and st["tri_hits"] == ost["tri_hits"]
    assert st["nodes_closest"] == ost["nodes_closest"] and st["nodes_any"] == ost["nodes_any"]
    assert st["nee_evals"] == ost["nee_evals"] and st["zero_radiance"] == ost["zero_radiance"]
    assert st["path_length"] == ost["path_length"]
    assert st["ms_extend"] > 0 and st["n_extend_launches"] == 6 * st["n_passes"]
    plain, _ = gpu_small.render()  # the uninstrumented kernels (what bench.py times)
    assert_bitwise(plain, ref, "film, uninstrumented kernels")


def test_film_c1_bitwise_vs_oracle_and_reference_pins(gpu_c1, scene_c1, oracle):
    """BASELINE config 0 (400x400, 8 spp) end to end."""
    film, st = gpu_c1.render(collect_stats=True)
    ref, ost = oracle.render(scene_c1)
    assert_bitwise(film, ref, "C1 film")
    plain, _ = gpu_c1.render()
    assert_bitwise(plain, ref, "C1 film, uninstrumented kernels")
    # ray counts of the device equal the portable-mode oracle's exactly ...
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    # ... and sit within a handful of flipped paths of the reference's own counts
    # (5,509,699 + 2,009,697; the libm-mode oracle reproduces them exactly)
    assert abs(st["closest_rays"] - 5509699) <= 64 and abs(st["shadow_rays"] - 2009697) <= 64
    rgb = scene_c1.film_to_rgb(film)
    mean = float(rgb.mean(dtype=np.float64))
    assert abs(mean - 2.2752316) / 2.2752316 < 1e-5  # image-mean tolerance stated in SURVEY.md §7


def test_film_c1_vs_libm_oracle_tolerance(gpu_c1, scene_c1, oracle):
    """Against the oracle evaluated with the host libm (the mode pinned to the
    reference): <= 1e-4 relative on >= 99.9 % of pixels, image mean within 1e-5."""
    film, _ = gpu_c1.render()
    ref, _ = oracle.render(scene_c1, trig_mode=ob.TRIG_LIBM)
    a, b = scene_c1.film_to_rgb(film), scene_c1.film_to_rgb(ref)
    rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-6)
    frac_bad = float((rel.max(axis=2) > 1e-4).mean())
    assert frac_bad <= 1e-3, frac_bad
    assert abs(a.mean(dtype=np.float64) - b.mean(dtype=np.float64)) / b.mean(dtype=np.float64) < 1e-5


def test_furnace_scene_on_device(binding, oracle):
    """Analytic furnace scene of src/tests/analytic_scenes.cpp: exercises shading ON the
    emitting sphere, light sampling from inside the sphere and maxdepth 8 (Russian roulette
    on several bounces). Mean radiance 1.0 +- 0.02, and bitwise equal to the oracle."""
    import os
    scene = binding.HostScene(path=os.path.join(os.path.dirname(__file__), "golden", "scenes", "furnace_area.pbrt"))
    gpu = binding.GpuScene(scene)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(scene)
    assert_bitwise(film, ref, "furnace film")
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "furnace film, uninstrumented kernels")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["path_length"] == ost["path_length"]
    assert abs(float(scene.film_to_rgb(film).mean(dtype=np.float64)) - 1.0) < 0.02


def test_passes_and_sample_split_are_equivalent(gpu_small, scene_small):
    """The film is independent of how samples are chunked into wavefront passes."""
    full, _ = gpu_small.render()
    chunked, st = gpu_small.render(spp_per_pass=1)
    assert st["n_passes"] == 4
    assert_bitwise(chunked, full, "1 spp per pass vs one pass")


def test_tile_sharding_sums_to_full_film(gpu_small, scene_small, oracle):
    """Multi-GPU decomposition: rank r renders tiles t % n == r into a
    full-resolution film; the sum over ranks equals the single-GPU film."""
    full, _ = gpu_small.render()
    for nranks in (2, 3, 8):
        acc = np.zeros_like(full)
        for r in range(nranks):
            part, _ = gpu_small.render(tile_rank=r, tile_nranks=nranks)
            ref_part, _ = oracle.render(scene_small, tile_rank=r, tile_nranks=nranks)
            assert_bitwise(part, ref_part, f"shard {r}/{nranks}")
            acc += part
        # disjoint tiles: x + 0 is exact; only the k=0 halo splats add two non-zeros
        assert np.allclose(acc, full, rtol=1e-6, atol=0)
        assert np.array_equal(acc[..., 3], full[..., 3])


def test_film_on_device_pointer(gpu_small, scene_small):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    h, w = scene_small.film_shape
    film_t = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _, st = gpu_small.render(film_device_ptr=film_t.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    host, _ = gpu_small.render()
    assert_bitwise(film_t.cpu().numpy(), host, "device-resident film")


def test_cpp_cli_renders_same_film(scene_small, gpu_small, tmp_path):
    """The C++ host (`iile_pbrt`, GpuPathIntegrator::Render) produces the image the C ABI does."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pbrt-v3-iile_amd", "lib",
                       "iile_pbrt")
    out = tmp_path / "cli.pfm"
    p = subprocess.run([exe, os.path.join(os.path.dirname(exe), "..", "..", "scenes", "killeroo-simple.pbrt"),
                        "--xres", "160", "--yres", "120", "--spp", "4", "--outfile", str(out), "--stats"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    raw = out.read_bytes()
    head = b"PF\n160 120\n-1.0\n"
    assert raw.startswith(head)
    img = np.frombuffer(raw[len(head):], "<f4").reshape(120, 160, 3)[::-1]
    film, _ = gpu_small.render()
    assert_bitwise(img, scene_small.film_to_rgb(film), "CLI image")


def test_cpp_cli_ranked_branch_world1(scene_small, gpu_small, tmp_path):
    """`iile_pbrt --gpurank 0/1 --rendezvous F` goes through GpuPathIntegrator's communicator branch (device film,
    iile_dist_all_ok, iile_dist_film_reduce over RCCL, download, iile_dist_sum_u64 / max_f64; where Film::MergeFilmTile
    stands, film.cpp:135-148) with a communicator of one rank: its image equals the plain CLI's bit for bit, the summed
    statistics equal the single-rank ones, a stale rendezvous file from an earlier run is replaced, and the file is gone
    once the ranks have joined."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "pbrt-v3-iile_amd", "lib", "iile_pbrt")
    scene = os.path.join(repo, "scenes", "killeroo-simple.pbrt")
    size = ["--xres", "160", "--yres", "120", "--spp", "4", "--stats"]
    plain, ranked, rv = tmp_path / "plain.pfm", tmp_path / "ranked.pfm", tmp_path / "rendezvous"
    rv.write_bytes(bytes(128))  # what a round-2 run would have left behind
    p0 = subprocess.run([exe, scene, *size, "--outfile", str(plain)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p0.returncode == 0, p0.stdout
    p1 = subprocess.run([exe, scene, *size, "--outfile", str(ranked), "--gpurank", "0/1", "--rendezvous", str(rv), "--job", "77"],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p1.returncode == 0, p1.stdout
    assert plain.read_bytes() == ranked.read_bytes()
    assert not rv.exists()
    stats = [[l for l in p.stdout.splitlines() if l.startswith("rays:")] for p in (p0, p1)]
    assert stats[0] and stats[0] == stats[1], (p0.stdout, p1.stdout)
    # and the film is the C ABI's
    raw = ranked.read_bytes()
    head = b"PF\n160 120\n-1.0\n"
    img = np.frombuffer(raw[len(head):], "<f4").reshape(120, 160, 3)[::-1]
    film, _ = gpu_small.render()
    assert_bitwise(img, scene_small.film_to_rgb(film), "ranked CLI image")
    # one process, the devices of the node: `--gpus 1` runs GpuPathIntegrator::RenderAllDevices — a host thread per device, an
    # in-process RCCL rendezvous, the same communicator branch — with one device: the same image and statistics again. (With
    # several GPUs visible that path is the default of `iile_pbrt scene.pbrt`; the one-GPU boxes of this pool take the plain one.)
    allp = tmp_path / "all.pfm"
    p3 = subprocess.run([exe, scene, *size, "--outfile", str(allp), "--gpus", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p3.returncode == 0, p3.stdout
    assert allp.read_bytes() == plain.read_bytes()
    assert [l for l in p3.stdout.splitlines() if l.startswith("rays:")] == stats[0], p3.stdout
    # a rank whose scene does not load leaves through the status exchange with an error, not a hang
    p2 = subprocess.run([exe, str(tmp_path / "missing.pbrt"), "--gpurank", "0/1", "--rendezvous", str(rv)],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p2.returncode == 1 and "Error" in p2.stdout


def test_boxroom_deep_bvh_bitwise(binding, oracle, tmp_path):
    """Synthetic closed room (tests/boxroom.py; SURVEY.md §8d's stand-in for the deep-BVH config):
    ~65 reference node visits per ray instead of killeroo-simple's 17, every path runs to
    maxdepth, twelve materials. Film and all counters bitwise equal to the oracle, with the
    instrumented (binary steps) and the plain (four-wide steps) kernels."""
    import boxroom
    path = tmp_path / "boxroom.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4))
    scene = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(scene)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(scene)
    assert_bitwise(film, ref, "boxroom film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["nodes_closest"] == ost["nodes_closest"] and st["nodes_any"] == ost["nodes_any"]
    assert st["tri_tests"] == ost["tri_tests"] and st["path_length"] == ost["path_length"]
    assert st["nodes_closest"] / st["closest_rays"] > 40  # it is a deep tree
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "boxroom film, uninstrumented kernels")
    # several samples per pass boundary and sharding on the same scene
    part0, _ = gpu.render(tile_rank=0, tile_nranks=2, spp_per_pass=3)
    part1, _ = gpu.render(tile_rank=1, tile_nranks=2, spp_per_pass=1)
    assert np.array_equal((part0 + part1)[..., 3], ref[..., 3])
    assert np.allclose(part0 + part1, ref, rtol=1e-6, atol=0)


def test_point_light_scenes_bitwise(binding, oracle, tmp_path):
    """Delta lights (first step): the reference's analytic point-light furnace
    scene and the synthetic box room lit by a point, a spot and a distant light, all bitwise
    equal to the oracle. (Only the point light has a test in the reference.)"""
    import os
    import boxroom
    furnace = binding.HostScene(path=os.path.join(os.path.dirname(__file__), "golden", "scenes", "furnace_point.pbrt"))
    path = tmp_path / "boxroom_point.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, light="point"))
    room = binding.HostScene(path=str(path))
    scenes = [("furnace", furnace), ("boxroom", room)]
    for kind in ("spot", "distant"):
        p2 = tmp_path / f"boxroom_{kind}.pbrt"
        p2.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, light=kind))
        scenes.append((f"boxroom {kind}", binding.HostScene(path=str(p2))))
    for name, scene in scenes:
        gpu = binding.GpuScene(scene)
        film, st = gpu.render(collect_stats=True)
        ref, ost = oracle.render(scene)
        assert float(scene.film_to_rgb(ref).mean()) > 1e-3, f"{name}: the light reaches nothing"
        assert_bitwise(film, ref, f"{name} (point light) film")
        assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
        assert st["path_length"] == ost["path_length"] and st["zero_radiance"] == ost["zero_radiance"]
        plain, _ = gpu.render()
        assert_bitwise(plain, ref, f"{name} (point light) film, uninstrumented kernels")


def test_specular_materials_bitwise(binding, oracle, tmp_path):
    """UberMaterial and MirrorMaterial: the reference's analytic uber scene, and
    the box room with uber / mirror / plastic / matte blobs under the area light, where paths
    reach the emitter through specular bounces (the `specularBounce` branch of Li)."""
    import os
    import boxroom
    furnace = binding.HostScene(path=os.path.join(os.path.dirname(__file__), "golden", "scenes", "furnace_uber.pbrt"))
    path = tmp_path / "boxroom_mixed.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, materials="mixed"))
    room = binding.HostScene(path=str(path))
    for name, scene in (("furnace", furnace), ("boxroom", room)):
        gpu = binding.GpuScene(scene)
        film, st = gpu.render(collect_stats=True)
        ref, ost = oracle.render(scene)
        assert_bitwise(film, ref, f"{name} (specular materials) film")
        assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
        assert st["nee_evals"] == ost["nee_evals"] and st["path_length"] == ost["path_length"]
        plain, _ = gpu.render()
        assert_bitwise(plain, ref, f"{name} (specular materials) film, uninstrumented kernels")


def test_uber_transmission_bitwise(binding, oracle, tmp_path):
    """UberMaterial's two SpecularTransmission lobes (uber.cpp:53-61, 94-99; refused until round 6): the pass-through of a surface
    that is not opaque — grey and coloured opacities — and the Kt lobe, alone and beside the diffuse / glossy / mirror lobes, on
    blobs of the box room at maxdepth 8 (BSDF::eta is 1 with the pass-through, the material's index without: etaScale and the
    roulette see the difference). Film and every counter against the oracle bit for bit, both kernel sets; the oracle's lobes are
    pinned by tests/test_oracle_pins.py::test_uber_transmission_pins. The IISPT runner's stages and the direct pass likewise."""
    import boxroom
    path = tmp_path / "boxroom_ubertrans.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, materials="ubertrans", maxdepth=8))
    room = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(room)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(room)
    assert_bitwise(film, ref, "uber transmission film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["nee_evals"] == ost["nee_evals"] and st["path_length"] == ost["path_length"]
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "uber transmission film, uninstrumented kernels")
    assert sum(ost["path_length"][5:]) > 0   # paths do get past bounce 4: the roulette ran
    # the IISPT stages on the same room. The direct pass: DirectProgressiveIntegrator::Li recurses through BOTH specular
    # transmissions of such a BSDF's vertex — SpecularTransmit's u[0] picks one of the two lobes, pdf 1/2 — walked per pixel
    # (k_direct_tree<.., 2>); the runner's first-intersection search follows whichever specular lobe Sample_f picked
    direct = gpu.render_direct(3)
    ref_direct = oracle.iispt_direct(room, 3)
    assert np.array_equal(direct.view(np.uint64), ref_direct.view(np.uint64))
    assert (direct[..., :3] > 0).any()
    task = binding.IisptTask(0, 0, 96, 64, 8, 0, 0)
    valid, pos, dr = gpu.iispt_hemi_points(task)
    rv, rp, rd = oracle.iispt_hemi_points(room, task)
    assert np.array_equal(valid, rv) and np.array_equal(pos.view(np.uint32), rp.view(np.uint32)) and np.array_equal(dr.view(np.uint32), rd.view(np.uint32))
    nn = np.random.default_rng(4).uniform(0.0, 3.0, valid.shape + (32, 32, 3)).astype(np.float32)
    out = gpu.iispt_gather(task, valid, pos, dr, nn)
    assert np.array_equal(out.view(np.uint32), oracle.iispt_gather(room, task, valid, pos, dr, nn).view(np.uint32)) and (out[..., 3] == 0.5).sum() > 500
    # "opacity" and Kt as image textures (uber.cpp:117, 53: the pass-through varies over the surface), in the textured room: the film
    # and counters, the direct pass (differentials through the transmissions), hemi points
    path = tmp_path / "boxroom_ubertrans_tex.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, materials="ubertrans", maxdepth=8, n_blobs=12, ico_levels=3, textures=str(tmp_path / "tex")))
    room = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(room)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(room)
    assert_bitwise(film, ref, "uber with opacity / Kt images: film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"] and st["path_length"] == ost["path_length"]
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "uber with opacity / Kt images: film, uninstrumented kernels")
    direct = gpu.render_direct(2)
    assert np.array_equal(direct.view(np.uint64), oracle.iispt_direct(room, 2).view(np.uint64))
    valid, pos, dr = gpu.iispt_hemi_points(task)
    rv, rp, rd = oracle.iispt_hemi_points(room, task)
    assert np.array_equal(valid, rv) and np.array_equal(pos.view(np.uint32), rp.view(np.uint32)) and np.array_equal(dr.view(np.uint32), rd.view(np.uint32))


def test_rough_glass_bitwise(binding, oracle, tmp_path):
    """GlassMaterial with uroughness = vroughness != 0 (glass.cpp:66-90; refused until round 6): MicrofacetReflection +
    MicrofacetTransmission — glossy lobes on both sides of the surface, so EstimateDirect lights a point THROUGH the surface and
    BSDF::f / Pdf / Sample_f carry a transmission term. The lobes in the canonical frame for every material kind the loader makes
    (f, pdf, Sample_f over the whole sphere of directions), then the box room with rough refractive blobs at maxdepth 8: film and
    every counter against the oracle bit for bit, both kernel sets. The oracle's lobes are pinned by
    tests/test_oracle_pins.py::test_rough_glass_pins. The IISPT runner's stages and the direct pass on the same room likewise."""
    import boxroom
    mats = tmp_path / "mats.pbrt"
    mats.write_text('''Camera "perspective"
Film "image" "integer xresolution" [4] "integer yresolution" [4]
Sampler "halton" "integer pixelsamples" [1]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "color L" [1 1 1]
  Shape "sphere" "float radius" [1]
AttributeEnd
Material "glass" "color Kr" [.9 .9 .9] "color Kt" [.8 .9 1] "float uroughness" [.2] "float vroughness" [.2] "float index" [1.5]
Shape "trianglemesh" "point P" [0 0 5 1 0 5 0 1 5] "integer indices" [0 1 2]
Material "glass" "color Kr" [0 0 0] "float uroughness" [.05] "float vroughness" [.05] "bool remaproughness" ["false"] "float index" [1.33]
Shape "trianglemesh" "point P" [0 0 6 1 0 6 0 1 6] "integer indices" [0 1 2]
Material "glass" "color Kt" [0 0 0] "float uroughness" [.6] "float vroughness" [.6] "float index" [1.7]
Shape "trianglemesh" "point P" [0 0 7 1 0 7 0 1 7] "integer indices" [0 1 2]
Material "uber" "color Kd" [.25 .3 .2] "color Ks" [.3 .3 .3] "color Kr" [.2 .2 .2] "color Kt" [.3 .3 .3] "color opacity" [.7 .6 .5] "float roughness" [.2]
Shape "trianglemesh" "point P" [0 0 8 1 0 8 0 1 8] "integer indices" [0 1 2]
Material "matte" "color Kd" [.6 .5 .4] "float sigma" [30]
Shape "trianglemesh" "point P" [0 0 9 1 0 9 0 1 9] "integer indices" [0 1 2]
WorldEnd
''')
    scene = binding.HostScene(path=str(mats))
    gpu = binding.GpuScene(scene)
    rng = np.random.default_rng(15)
    n = 4096

    def dirs(m):
        v = rng.normal(size=(m, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v.astype(np.float32)

    wo, wi = dirs(n), dirs(n)
    wo[:8, 2] = [0, 1e-8, -1e-8, 1, -1, 0.99995, 0.5, -0.5]
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    u[:4] = [[0, 0], [0.99999994, 0.99999994], [0.5, 0.5], [0.49999997, 0.5]]
    assert scene.info["n_materials"] >= 5
    for mat in range(scene.info["n_materials"]):
        ev = gpu.bsdf_eval(mat, wo, wi)
        assert_bitwise(ev, oracle.bsdf_eval(scene, mat, wo, wi), f"BSDF f/pdf material {mat}")
        sm = gpu.bsdf_sample(mat, wo, u)
        assert_bitwise(sm, oracle.bsdf_sample(scene, mat, wo, u), f"BSDF Sample_f material {mat}")
        if mat in (1, 2):   # (material 0 is the light's default matte)
            assert (ev[wo[:, 2] * wi[:, 2] < 0][:, :3] > 0).any()   # the transmission lobe is there
    path = tmp_path / "boxroom_roughglass.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, materials="roughglass", maxdepth=8))
    room = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(room)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(room)
    assert_bitwise(film, ref, "rough glass film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["nee_evals"] == ost["nee_evals"] and st["path_length"] == ost["path_length"]
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "rough glass film, uninstrumented kernels")
    # the IISPT stages on the same room (the runner's first-intersection search, the gather over predicted hemispheres, the direct pass
    # — per-pixel tree walk, as for every scene with glass): the rough lobes are glossy, so Li does not recurse through them
    direct = gpu.render_direct(2)
    assert np.array_equal(direct.view(np.uint64), oracle.iispt_direct(room, 2).view(np.uint64))
    task = binding.IisptTask(0, 0, 96, 64, 8, 0, 0)
    valid, pos, dr = gpu.iispt_hemi_points(task)
    rv, rp, rd = oracle.iispt_hemi_points(room, task)
    assert np.array_equal(valid, rv) and np.array_equal(pos.view(np.uint32), rp.view(np.uint32)) and np.array_equal(dr.view(np.uint32), rd.view(np.uint32))
    nn = np.random.default_rng(3).uniform(0.0, 3.0, valid.shape + (32, 32, 3)).astype(np.float32)
    out = gpu.iispt_gather(task, valid, pos, dr, nn)
    assert np.array_equal(out.view(np.uint32), oracle.iispt_gather(room, task, valid, pos, dr, nn).view(np.uint32)) and (out[..., 3] == 0.5).sum() > 500


def test_anisotropic_roughness_bitwise(binding, oracle, tmp_path):
    """uroughness != vroughness on uber and glass (uber.cpp:73-86, glass.cpp:52-73; refused until round 6): an anisotropic
    TrowbridgeReitzDistribution(alphax, alphay) in D, Lambda and the visible-normal sampling. BSDF f / pdf / Sample_f in the canonical
    frame against the oracle bit for bit, then the box room with such blobs at maxdepth 8 — film and counters, both kernel sets — and
    the IISPT stages on it. Pins: tests/test_oracle_pins.py::test_anisotropic_roughness_pins and the reference's own
    BSDFSampling.TR_VA_0p3_0p15 (test_bsdf_sampling_chi_square)."""
    import boxroom
    mats = tmp_path / "mats.pbrt"
    mats.write_text('''Camera "perspective"
Film "image" "integer xresolution" [4] "integer yresolution" [4]
Sampler "halton" "integer pixelsamples" [1]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "color L" [1 1 1]
  Shape "sphere" "float radius" [1]
AttributeEnd
Material "uber" "color Kd" [.2 .3 .4] "color Ks" [.6 .5 .4] "color Kr" [.1 .1 .1] "float uroughness" [.3] "float vroughness" [.15]
Shape "trianglemesh" "point P" [0 0 5 1 0 5 0 1 5] "integer indices" [0 1 2]
Material "uber" "color Kd" [0 0 0] "color Ks" [1 1 1] "float uroughness" [.02] "float vroughness" [.6] "bool remaproughness" ["false"] "color opacity" [.8 .7 .6]
Shape "trianglemesh" "point P" [0 0 6 1 0 6 0 1 6] "integer indices" [0 1 2]
Material "glass" "float uroughness" [0] "float vroughness" [.3] "float index" [1.5]
Shape "trianglemesh" "point P" [0 0 7 1 0 7 0 1 7] "integer indices" [0 1 2]
Material "glass" "color Kr" [.9 .8 .7] "color Kt" [.7 .8 .9] "float uroughness" [.4] "float vroughness" [.05] "bool remaproughness" ["false"] "float index" [1.33]
Shape "trianglemesh" "point P" [0 0 8 1 0 8 0 1 8] "integer indices" [0 1 2]
WorldEnd
''')
    scene = binding.HostScene(path=str(mats))
    gpu = binding.GpuScene(scene)
    rng = np.random.default_rng(16)
    n = 4096

    def dirs(m):
        v = rng.normal(size=(m, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v.astype(np.float32)

    wo, wi = dirs(n), dirs(n)
    wo[:8, 2] = [0, 1e-8, -1e-8, 1, -1, 0.99995, 0.5, -0.5]
    u = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    u[:4] = [[0, 0], [0.99999994, 0.99999994], [0.5, 0.5], [0.49999997, 0.5]]
    assert scene.info["n_materials"] >= 5
    for mat in range(scene.info["n_materials"]):
        ev = gpu.bsdf_eval(mat, wo, wi)
        assert_bitwise(ev, oracle.bsdf_eval(scene, mat, wo, wi), f"BSDF f/pdf material {mat}")
        assert_bitwise(gpu.bsdf_sample(mat, wo, u), oracle.bsdf_sample(scene, mat, wo, u), f"BSDF Sample_f material {mat}")
    # anisotropy is really there: swapping x and y of both directions changes f for material 1 (the light's default matte is material 0)
    sw = lambda v: np.ascontiguousarray(v[:, [1, 0, 2]])
    assert not np.array_equal(gpu.bsdf_eval(1, wo, wi)[:, :3], gpu.bsdf_eval(1, sw(wo), sw(wi))[:, :3])
    path = tmp_path / "boxroom_aniso.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, materials="aniso", maxdepth=8, n_blobs=9))
    room = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(room)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(room)
    assert_bitwise(film, ref, "anisotropic roughness film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["nee_evals"] == ost["nee_evals"] and st["path_length"] == ost["path_length"]
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "anisotropic roughness film, uninstrumented kernels")
    direct = gpu.render_direct(2)
    assert np.array_equal(direct.view(np.uint64), oracle.iispt_direct(room, 2).view(np.uint64))
    task = binding.IisptTask(0, 0, 96, 64, 8, 0, 0)
    valid, pos, dr = gpu.iispt_hemi_points(task)
    rv, rp, rd = oracle.iispt_hemi_points(room, task)
    assert np.array_equal(valid, rv) and np.array_equal(pos.view(np.uint32), rp.view(np.uint32)) and np.array_equal(dr.view(np.uint32), rd.view(np.uint32))
    nn = np.random.default_rng(5).uniform(0.0, 3.0, valid.shape + (32, 32, 3)).astype(np.float32)
    out = gpu.iispt_gather(task, valid, pos, dr, nn)
    assert np.array_equal(out.view(np.uint32), oracle.iispt_gather(room, task, valid, pos, dr, nn).view(np.uint32)) and (out[..., 3] == 0.5).sum() > 500
    # float images for "uroughness" / "vroughness" / "roughness" (uber.cpp:73-86: u from an image and v a number, both from images, "roughness"
    # ignored beside "uroughness", v following u's image), in the textured room
    path = tmp_path / "boxroom_aniso_tex.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, materials="aniso", maxdepth=8, n_blobs=12, ico_levels=3, textures=str(tmp_path / "tex")))
    room = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(room)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(room)
    assert_bitwise(film, ref, "roughness images per axis: film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"] and st["path_length"] == ost["path_length"]
    assert_bitwise(gpu.render()[0], ref, "roughness images per axis: film, uninstrumented kernels")
    assert np.array_equal(gpu.render_direct(2).view(np.uint64), oracle.iispt_direct(room, 2).view(np.uint64))


def test_glass_scenes_bitwise(binding, oracle, tmp_path):
    """GlassMaterial (FresnelSpecular: specular reflection + transmission, the etaScale branch of Li
    and of its Russian roulette). No test of the reference covers glass; the restatement is checked
    by the white-furnace property (tests/golden/scenes/furnace_glass.pbrt: radiance 1 through a
    lossless glass ball) and the device against the oracle bit for bit, on that scene and on the box
    room with refractive blobs at maxdepth 8 (Russian roulette with etaScale != 1)."""
    import os
    import boxroom
    furnace = binding.HostScene(path=os.path.join(os.path.dirname(__file__), "golden", "scenes", "furnace_glass.pbrt"))
    path = tmp_path / "boxroom_glass.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, materials="glass", maxdepth=8))
    room = binding.HostScene(path=str(path))
    for name, scene in (("furnace", furnace), ("boxroom", room)):
        gpu = binding.GpuScene(scene)
        film, st = gpu.render(collect_stats=True)
        ref, ost = oracle.render(scene)
        assert_bitwise(film, ref, f"{name} (glass) film")
        assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
        assert st["nee_evals"] == ost["nee_evals"] and st["path_length"] == ost["path_length"]
        plain, _ = gpu.render()
        assert_bitwise(plain, ref, f"{name} (glass) film, uninstrumented kernels")
        if name == "furnace":
            assert abs(float(scene.film_to_rgb(film).mean(dtype=np.float64)) - 1.0) < 0.02


def test_several_lights_bitwise(binding, oracle, tmp_path):
    """More than one light: the spatial light distribution (tabulated per voxel on the device at
    scene creation) picks the light to sample. The reference's four-point-light analytic scene, and
    the box room lit by an emitting sphere, a point light and a spot light at once."""
    import os
    import boxroom
    furnace = binding.HostScene(path=os.path.join(os.path.dirname(__file__), "golden", "scenes", "furnace_4points.pbrt"))
    path = tmp_path / "boxroom_multi.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, light="multi", materials="mixed"))
    room = binding.HostScene(path=str(path))
    assert room.info["n_lights"] == 3
    # triangle emitters: the closed tetrahedron furnace, and a two-triangle ceiling panel + a point light
    tetra = binding.HostScene(path=os.path.join(os.path.dirname(__file__), "golden", "scenes", "furnace_tetrahedron.pbrt"))
    path2 = tmp_path / "boxroom_quad.pbrt"
    path2.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, light="quad"))
    panel = binding.HostScene(path=str(path2))
    assert panel.info["n_lights"] == 3
    # the other light sample strategies (path.cpp:231): one Distribution1D for the whole scene
    others = []
    for strategy in ("uniform", "power"):
        p3 = tmp_path / f"boxroom_multi_{strategy}.pbrt"
        p3.write_text(path.read_text().replace('Integrator "path"', 'Integrator "path" "string lightsamplestrategy" ["%s"]' % strategy))
        others.append((f"boxroom, {strategy} light sampling", binding.HostScene(path=str(p3))))
    for name, scene in [("furnace", furnace), ("boxroom", room), ("tetrahedron", tetra), ("panel", panel)] + others:
        gpu = binding.GpuScene(scene)
        film, st = gpu.render(collect_stats=True)
        ref, ost = oracle.render(scene)
        assert float(scene.film_to_rgb(ref).mean()) > 1e-3
        assert_bitwise(film, ref, f"{name} (several lights) film")
        assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
        assert st["tri_tests"] == ost["tri_tests"] and st["tri_hits"] == ost["tri_hits"]
        assert st["nee_evals"] == ost["nee_evals"] and st["zero_radiance"] == ost["zero_radiance"]
        plain, _ = gpu.render()
        assert_bitwise(plain, ref, f"{name} (several lights) film, uninstrumented kernels")


SPHERES_SCENE = """LookAt 0 -8 3  0 0 1  0 0 1
Camera "perspective" "float fov" [45]
Film "image" "integer xresolution" [96] "integer yresolution" [64]
Sampler "halton" "integer pixelsamples" [4]
Integrator "path" "integer maxdepth" [5]
WorldBegin
AttributeBegin
  Translate 1 -1 4
  Rotate 30 0 1 0
  AreaLightSource "area" "color L" [30 30 30]
  Shape "sphere" "float radius" [0.5]
AttributeEnd
AttributeBegin
  Material "plastic" "color Kd" [.2 .5 .3] "color Ks" [.3 .3 .3] "float roughness" [0.1]
  Translate -1.5 0 1
  Rotate 40 1 1 0
  Scale 1 0.6 1.3
  Shape "sphere" "float radius" [0.8]
AttributeEnd
AttributeBegin
  Material "matte" "color Kd" [.6 .3 .2]
  Translate 1.2 0.5 0.7
  Shape "sphere" "float radius" [0.7]
AttributeEnd
AttributeBegin
  Material "matte" "color Kd" [.5 .5 .5]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-5 -5 0  5 -5 0  5 5 0  -5 5 0]
AttributeEnd
WorldEnd
"""


def test_several_spheres_with_transforms_bitwise(binding, oracle, tmp_path):
    """Three spheres under different transforms (a rotated emitter, a rotated and unevenly scaled plastic one, a translated
    matte one) over a floor: the traversal's leaf test takes the spheres of a wavefront's lanes one at a time with the sphere's
    fields in scalar registers, k_shade reads the light's sphere the same way, and the sphere's centre is worked out at
    upload — all three against the oracle, which does none of that."""
    path = tmp_path / "spheres.pbrt"
    path.write_text(SPHERES_SCENE)
    scene = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(scene)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(scene)
    assert float(scene.film_to_rgb(ref).mean()) > 1e-3
    assert_bitwise(film, ref, "three spheres film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["sphere_tests"] == ost["sphere_tests"] and st["nee_evals"] == ost["nee_evals"]
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "three spheres film, uninstrumented kernels")


PARTIAL_SPHERES_SCENE = """LookAt 0 -8 3  0 0 1  0 0 1
Camera "perspective" "float fov" [45]
Film "image" "integer xresolution" [96] "integer yresolution" [64]
Sampler "halton" "integer pixelsamples" [4]
Integrator "path" "integer maxdepth" [5]
WorldBegin
AttributeBegin
  Translate 1 -1 4
  Rotate 30 0 1 0
  AreaLightSource "area" "color L" [30 30 30]
  Shape "sphere" "float radius" [0.5] "float zmin" [-0.4] "float phimax" [300]
AttributeEnd
Texture "checks" "spectrum" "imagemap" "string filename" ["checks.pfm"]
Texture "bumps" "float" "imagemap" "string filename" ["bumps.pfm"]
AttributeBegin
  Material "plastic" "texture Kd" ["checks"] "color Ks" [.3 .3 .3] "float roughness" [0.1] "texture bumpmap" ["bumps"]
  Translate -1.5 0 1
  Rotate 40 1 1 0
  Scale 1 0.6 1.3
  Shape "sphere" "float radius" [0.8]
AttributeEnd
AttributeBegin
  Material "matte" "texture Kd" ["checks"]
  Translate 1.2 0.5 0.9
  Rotate -25 1 0 0
  Shape "sphere" "float radius" [0.9] "float zmax" [0.35] "float phimax" [250]
AttributeEnd
AttributeBegin
  Material "mirror" "color Kr" [.8 .8 .8]
  Translate 0 1.8 1.2
  Rotate 70 0 1 0
  Shape "sphere" "float radius" [1.1] "float zmin" [-0.3] "float zmax" [0.8] "float phimax" [200]
AttributeEnd
AttributeBegin
  Material "matte" "color Kd" [.5 .5 .5]
  Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-5 -5 0  5 -5 0  5 5 0  -5 5 0]
AttributeEnd
WorldEnd
"""


def test_partial_and_textured_spheres_bitwise(binding, oracle, tmp_path):
    """Sphere::Intersect's clipping (zmin / zmax / phimax: sphere.cpp:89-104, the second root tried when the first is cut away) and a
    sphere hit's (u, v), dp/du, dp/dv, dn/du, dn/dv for image textures and bump maps (:107-143) — both refused on the device until
    round 6: a cut emitter (its Sample still draws from the whole sphere, as the reference's does), a textured and bump-mapped
    ellipsoid, a textured bowl and a mirror band over a floor; closest / any hits of random rays, then film and every counter
    against the oracle bit for bit, both kernel sets. Pins of the oracle: tests/test_oracle_pins.py::test_partial_and_textured_spheres_pins."""
    rng = np.random.default_rng(21)
    chk = np.zeros((8, 8, 3), np.float32)
    chk[::2, ::2] = chk[1::2, 1::2] = (.8, .3, .2)
    chk[::2, 1::2] = chk[1::2, ::2] = (.2, .4, .8)
    (tmp_path / "checks.pfm").write_bytes(b"PF\n8 8\n-1.0\n" + chk.tobytes())
    bmp = np.repeat(rng.uniform(0, .05, (16, 16, 1)).astype(np.float32), 3, 2)
    (tmp_path / "bumps.pfm").write_bytes(b"PF\n16 16\n-1.0\n" + bmp.tobytes())
    path = tmp_path / "partial_spheres.pbrt"
    path.write_text(PARTIAL_SPHERES_SCENE)
    scene = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(scene)
    # kernel level: rays at and around the spheres
    n = 20000
    o = rng.uniform((-4, -6, 0.05), (4, 4, 5), (n, 3)).astype(np.float32)
    tgt = rng.uniform((-3, -2, 0), (3, 3, 4.5), (n, 3)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.float32(np.inf), rng.uniform(0.5, 2.0, n).astype(np.float32)).astype(np.float32)
    for instrumented in (True, False):
        prim, tb, _ = gpu.trace_closest(o, d, tmax, instrumented=instrumented)
        rprim, rtb = oracle.intersect(scene, o, d, tmax)
        assert np.array_equal(prim, rprim) and (prim >= 0).mean() > 0.3
        assert_bitwise(tb[prim >= 0], rtb[prim >= 0], "partial spheres closest hit")
        hit = gpu.trace_any(o, d, tmax, instrumented=instrumented)[0]
        assert np.array_equal(hit, oracle.intersect_p(scene, o, d, tmax))
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(scene)
    assert float(scene.film_to_rgb(ref).mean()) > 1e-3
    assert_bitwise(film, ref, "partial / textured spheres film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["sphere_tests"] == ost["sphere_tests"] and st["nee_evals"] == ost["nee_evals"] and st["path_length"] == ost["path_length"]
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "partial / textured spheres film, uninstrumented kernels")
    # the IISPT direct pass on the same scene (its kernels build the same interactions)
    direct = gpu.render_direct(2)
    assert np.array_equal(direct.view(np.uint64), oracle.iispt_direct(scene, 2).view(np.uint64))


@pytest.mark.parametrize("seed,light", [(1, "quad"), (2, "multi"), (3, "area"), (4, "spot")])
def test_random_rooms_bitwise(binding, oracle, tmp_path, seed, light):
    """Differently seeded box rooms (other blob positions, sizes, noise, material parameters) with all
    five material kinds at once, under each light set-up, at maxdepth 7: film and counters bitwise
    equal to the oracle with both kernel builds."""
    import boxroom
    path = tmp_path / "room.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=80, yres=56, spp=3, ico_levels=3, n_blobs=10, wall_n=12, seed=seed,
                                         maxdepth=7, light=light, materials="all"))
    scene = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(scene)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(scene)
    assert float(scene.film_to_rgb(ref).mean()) > 1e-3
    assert_bitwise(film, ref, f"room seed {seed} / {light}")
    for k_dev, k_ref in (("closest_rays", "regular_rays"), ("shadow_rays", "shadow_rays"), ("tri_tests", "tri_tests"),
                         ("nodes_closest", "nodes_closest"), ("nodes_any", "nodes_any"), ("nee_evals", "nee_evals"),
                         ("zero_radiance", "zero_radiance"), ("path_length", "path_length")):
        assert st[k_dev] == ost[k_ref], k_dev
    plain, _ = gpu.render(spp_per_pass=2)
    assert_bitwise(plain, ref, f"room seed {seed} / {light}, uninstrumented kernels, two passes")


def test_infinite_light_bitwise(binding, oracle, tmp_path):
    """InfiniteAreaLight: the white-furnace sky scene (constant), the box room open to a tinted sky
    with a point light inside (two lights: the spatial light distribution samples the sky too), and the
    textured room under an environment map (a 12 x 6 lat-long PFM resampled to 16 x 8, its 32 x 16
    Distribution2D searched on the device). The device evaluates SphericalPhi / SphericalTheta with the
    portable atan2 / acos of the oracle."""
    import os
    import boxroom
    sky = binding.HostScene(path=os.path.join(os.path.dirname(__file__), "golden", "scenes", "furnace_sky.pbrt"))
    path = tmp_path / "boxroom_sky.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, light="sky", materials="all", maxdepth=6))
    room = binding.HostScene(path=str(path))
    assert room.info["n_lights"] == 2
    path2 = tmp_path / "boxroom_env.pbrt"
    path2.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, light="envmap", materials="mixed", textures=str(tmp_path / "img")))
    env = binding.HostScene(path=str(path2))
    assert env.info["n_lights"] == 1 and env.texture(0)[1][0].shape == (8, 16, 3)  # Lmap: the first pyramid
    for name, scene in (("sky furnace", sky), ("boxroom sky", room), ("boxroom under an environment map", env)):
        gpu = binding.GpuScene(scene)
        film, st = gpu.render(collect_stats=True)
        ref, ost = oracle.render(scene)
        assert float(scene.film_to_rgb(ref).mean()) > 1e-2
        assert_bitwise(film, ref, f"{name} film")
        assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
        assert st["zero_radiance"] == ost["zero_radiance"] and st["path_length"] == ost["path_length"]
        plain, _ = gpu.render()
        assert_bitwise(plain, ref, f"{name} film, uninstrumented kernels")
    assert abs(float(sky.film_to_rgb(gpu_film := binding.GpuScene(sky).render()[0]).mean(dtype=np.float64)) - 1.0) < 0.005


def test_film_1080p_bitwise_vs_oracle(binding, oracle):
    """BASELINE config 1's frame (killeroo-simple 1920x1080) at 8 of its 64 pixel samples: the film of the
    timed kernels is bit for bit the oracle's (16.6 M camera samples, 95 M rays); with all 64 samples the
    GPU film must be the same whether rendered in one pass or in eight (the oracle would need minutes)."""
    scene = binding.HostScene(xres=1920, yres=1080, spp=8)
    gpu = binding.GpuScene(scene)
    film, _ = gpu.render()
    ref, ost = oracle.render(scene)
    assert ost["camera_rays"] == 1920 * 1080 * 8
    assert_bitwise(film, ref, "1080p x 8 spp film")
    scene64 = binding.HostScene(xres=1920, yres=1080, spp=64)
    gpu64 = binding.GpuScene(scene64)
    one, st1 = gpu64.render()
    eight, st8 = gpu64.render(spp_per_pass=8)
    import os
    if "IILE_WORKSPACE_MB" not in os.environ:  # (a small workspace splits both renders further)
        assert st1["n_passes"] == 1 and st8["n_passes"] == 8
    assert_bitwise(eight, one, "1080p x 64 spp, eight passes vs one")


def test_image_textures_bitwise(binding, oracle, tmp_path):
    """Image textures: ImageTexture::Evaluate on the device — UV mapping, EWA with the
    anisotropy clamp, trilinear, the bilinear fallback, repeat / clamp / black wrapping, the portable log2 of
    the level choice — bit for bit the oracle's on random lookups in all five textures of the textured room
    (a Lanczos-resampled PFM, an inverse-gamma PNG, a run-length TGA); then the room itself, where the camera
    rays' differentials are rebuilt in k_shade from the pixel: film and counters bitwise, with the instrumented
    and the plain kernels, in several passes and shards; and a thin-lens camera (the lens branch of
    GenerateRayDifferential)."""
    import boxroom
    path = tmp_path / "room_tex.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, textures=str(tmp_path / "img")))
    scene = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(scene)
    rng = np.random.default_rng(2024)
    n = 20000
    uv = rng.uniform(-1.5, 2.5, (n, 2)).astype(np.float32)
    duv = (rng.standard_normal((n, 4)) * 10.0 ** rng.uniform(-5, 0.3, (n, 1))).astype(np.float32)
    duv[::5] = 0            # no differentials: bilinear at level 0
    duv[1::9, 2:] = 0       # a degenerate ellipse (minor axis 0)
    duv[2::13, :2] *= 50    # anisotropy beyond maxanisotropy
    for tex in range(5):
        assert_bitwise(gpu.texture_eval(tex, uv, duv), oracle.texture_eval(scene, tex, uv, duv), f"texture {tex} lookups")
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(scene)
    assert_bitwise(film, ref, "textured room film")
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["path_length"] == ost["path_length"] and st["zero_radiance"] == ost["zero_radiance"]
    plain, _ = gpu.render()
    assert_bitwise(plain, ref, "textured room film, uninstrumented kernels")
    part0, _ = gpu.render(tile_rank=0, tile_nranks=2, spp_per_pass=3)
    part1, _ = gpu.render(tile_rank=1, tile_nranks=2, spp_per_pass=1)
    assert np.array_equal((part0 + part1)[..., 3], ref[..., 3])
    assert np.allclose(part0 + part1, ref, rtol=1e-6, atol=0)
    # the textures must matter: the same room without them renders differently
    path2 = tmp_path / "room_plain.pbrt"
    path2.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4))
    other, _ = binding.GpuScene(binding.HostScene(path=str(path2))).render()
    assert np.abs(other[..., :3] - film[..., :3]).max() > 0.01
    # thin lens
    lens = path.read_text().replace('Camera "perspective" "float fov" [55]',
                                    'Camera "perspective" "float fov" [55] "float lensradius" [.15] "float focaldistance" [9]')
    assert lens != path.read_text()
    path3 = tmp_path / "room_tex_lens.pbrt"
    path3.write_text(lens)
    scene3 = binding.HostScene(path=str(path3))
    film3, _ = binding.GpuScene(scene3).render()
    ref3, _ = oracle.render(scene3)
    assert_bitwise(film3, ref3, "textured room through a thin lens")
    assert not np.array_equal(film3, film)


def test_wide_pixel_filters_bitwise(binding, oracle, tmp_path):
    """Pixel filters wider than the one-pixel box (FilmTile::AddSample with the filter table, film.h:153-193): every
    filter of src/filters with default and non-default parameters, on the furnace (a border-dominated 12 x 10 image)
    and on the box room — the film of the sample store + gather kernels is bit for bit the oracle's (same sums in
    the same order: pixel-major inside a tile, tiles in index order), in one pass and in several, and shard by shard."""
    import boxroom
    from test_oracle_pins import FILTERS, _FILTER_SCENE
    for i, line in enumerate(FILTERS):
        path = tmp_path / f"filter{i}.pbrt"
        path.write_text(_FILTER_SCENE % (12, 10, line, 4))
        scene = binding.HostScene(path=str(path))
        gpu = binding.GpuScene(scene)
        ref, ost = oracle.render(scene)
        film, st = gpu.render(collect_stats=True)
        assert st["camera_rays"] == ost["camera_rays"]
        assert_bitwise(film, ref, f"{line}: film")
        assert_bitwise(gpu.render(spp_per_pass=3)[0], ref, f"{line}: film in two passes")
    for line in (FILTERS[0], FILTERS[3], FILTERS[4], 'Sampler "halton" "bool samplepixelcenter" ["true"] "integer pixelsamples" [3]\nPixelFilter "triangle"',
                 'Sampler "halton" "bool samplepixelcenter" ["true"] "integer pixelsamples" [2]'):
        path = tmp_path / "room_filter.pbrt"
        text = boxroom.boxroom_pbrt(xres=96, yres=64, spp=4)
        assert 'Sampler "halton"' in text
        if line.startswith("Sampler"):  # every sample through its pixel's centre (halton.cpp:119), box and wide filter
            text = text.replace('Sampler "halton" "integer pixelsamples" [4]', line)
            assert "samplepixelcenter" in text
            path.write_text(text)
        else:
            path.write_text(text.replace('Sampler "halton"', line + '\nSampler "halton"'))
        scene = binding.HostScene(path=str(path))
        gpu = binding.GpuScene(scene)
        ref, _ = oracle.render(scene)
        assert_bitwise(gpu.render()[0], ref, f"box room, {line}")
        for rank in range(3):  # shards: each rank's partial film is the oracle's for the same tiles
            part, _ = gpu.render(tile_rank=rank, tile_nranks=3, spp_per_pass=1 + rank)
            pref, _ = oracle.render(scene, tile_rank=rank, tile_nranks=3)
            assert_bitwise(part, pref, f"box room, {line}, shard {rank} of 3")


def test_iispt_probe_pass_bitwise(binding, oracle, tmp_path):
    """The IISPT probe pass: hemispheric cameras rendered by IISPTdIntegrator::RenderView — a
    32 x 32 film behind a Gaussian filter, one Halton sample per pixel, depth 3, no emitted light at the camera ray's
    own vertex — plus the first hits' camera-space normals and distances: the three inputs of the IISPT network.
    A batch of probes in one wavefront pass gives, probe by probe, the oracle's images bit for bit: on killeroo-simple,
    in the point-light and the sky furnaces, and in the box room with three lights and specular materials (probe
    directions include the +z / -z axes, where CreateHemisphericCamera switches its up vector)."""
    import os
    import boxroom
    gold = os.path.join(os.path.dirname(__file__), "golden", "scenes")
    rng = np.random.default_rng(77)

    def probes(n, lo, hi):
        pos = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
        d = rng.standard_normal((n, 3)).astype(np.float32)
        d[0] = (0, 0, 1)
        d[1] = (0, 0, -2.5)
        d[2] = (1e-3, 0, 1)
        return pos, d

    path = tmp_path / "room_multi.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=32, yres=32, spp=1, light="multi", materials="mixed"))
    path_tex = tmp_path / "room_tex.pbrt"   # image textures, bump maps, alpha masks: the probe camera's own ray differentials
    path_tex.write_text(boxroom.boxroom_pbrt(xres=32, yres=32, spp=1, textures=str(tmp_path / "img")))
    cases = [("killeroo-simple", binding.HostScene(xres=64, yres=64, spp=1), probes(6, (-150, -100, -130), (250, 150, 0))),
             ("point furnace", binding.HostScene(path=os.path.join(gold, "furnace_point.pbrt")), probes(5, -0.4, 0.4)),
             ("sky furnace", binding.HostScene(path=os.path.join(gold, "furnace_sky.pbrt")), probes(5, (-3, -3, 1.2), (3, 3, 3))),
             ("box room, three lights", binding.HostScene(path=str(path)), probes(8, (-8, -8, -2), (8, 8, 8))),
             ("textured box room", binding.HostScene(path=str(path_tex)), probes(8, (-8, -8, -2), (8, 8, 8)))]
    for name, scene, (pos, d) in cases:
        gpu = binding.GpuScene(scene)
        inten, nrm, dist, st = gpu.render_probes(pos, d)
        assert st["n_passes"] == 1 and st["n_paths"] == len(pos) * 1024
        for i in range(len(pos)):
            oi, on, od = oracle.render_probe(scene, pos[i], d[i])
            assert_bitwise(inten[i], oi, f"{name}: intensity of probe {i}")
            assert_bitwise(nrm[i], on, f"{name}: normals of probe {i}")
            assert_bitwise(dist[i], od, f"{name}: distances of probe {i}")
        assert np.isfinite(inten).all() and inten.max() > 0
        # the frame renders as before after a probe batch (shared workspace, the frame's own film and sampler)
        assert_bitwise(gpu.render()[0], oracle.render(scene)[0], f"{name}: frame after the probe batch")
    # batches split across passes give the same images
    os.environ["IILE_WORKSPACE_MB"] = "64"
    try:
        name, scene, (pos, d) = cases[0]
        many = (np.tile(pos, (40, 1)), np.tile(d, (40, 1)))
        gpu = binding.GpuScene(scene)
        i2, n2, d2, st = gpu.render_probes(*many)
        assert st["n_passes"] > 1
        ref = binding.GpuScene(scene).render_probes(pos, d)
    finally:
        del os.environ["IILE_WORKSPACE_MB"]
    for j in range(40):
        assert_bitwise(i2[6 * j:6 * j + 6], ref[0], "probe intensities across passes")
        assert_bitwise(d2[6 * j:6 * j + 6], ref[2], "probe distances across passes")


def test_other_bvh_split_methods_bitwise(binding, oracle):
    """Trees of the other BVHAccel split methods ("middle", "equal", "hlbvh": leaves of more than four primitives,
    a different shape of the upper tree): film and traversal counters bitwise equal to the oracle's walk of the same
    tree, with the instrumented (binary) and the plain (four-wide) kernels."""
    import os
    src = open(binding.DEFAULT_SCENE).read()
    for method in ("middle", "equal", "hlbvh"):
        path = os.path.join(os.path.dirname(binding.DEFAULT_SCENE), f"_killeroo_gpu_{method}.pbrt")
        open(path, "w").write(src.replace("WorldBegin", 'Accelerator "bvh" "string splitmethod" ["%s"] "integer maxnodeprims" [8]\nWorldBegin' % method, 1))
        try:
            scene = binding.HostScene(path=path, xres=128, yres=96, spp=2)
        finally:
            os.remove(path)
        gpu = binding.GpuScene(scene)
        film, st = gpu.render(collect_stats=True)
        ref, ost = oracle.render(scene)
        assert_bitwise(film, ref, f"{method}: film")
        assert st["nodes_closest"] == ost["nodes_closest"] and st["nodes_any"] == ost["nodes_any"]
        assert st["tri_tests"] == ost["tri_tests"] and st["closest_rays"] == ost["regular_rays"]
        assert_bitwise(gpu.render()[0], ref, f"{method}: film, uninstrumented kernels")


def test_whole_number_film_positions_bitwise(binding, oracle):
    """Where pixel coordinates pass 1024 a film position px + u rounds to a whole number for u below (or within) 6e-5
    of 0 (or 1), and FilmTile::AddSample (film.h:159-166) then adds the sample to two pixels: the left / upper
    neighbour when it rounds down — as for the exact zeros of the first Halton samples — the right / lower one when it
    rounds up (whose own samples come later in the tile: the sum's order changes). A 1900 x 24 strip of killeroo-simple
    at 48 samples per pixel holds dozens of each kind: the film must be the oracle's bit for bit, in one pass (the
    radiances are gathered from the pass buffer) and in several (the paths involved are rendered again), shard by
    shard, and with the instrumented kernels."""
    scene = binding.HostScene(xres=1900, yres=24, spp=48)
    gpu = binding.GpuScene(scene)
    ref, ost = oracle.render(scene)
    weights = ref[..., 3]
    assert (weights > 48).sum() > 100 and (weights > 48)[:, 1100:].sum() > 20   # pixels that received a neighbour's sample
    film, st = gpu.render()
    assert st["n_passes"] == 1
    assert_bitwise(film, ref, "strip, one pass")
    assert_bitwise(gpu.render(spp_per_pass=10)[0], ref, "strip, five passes")
    assert_bitwise(gpu.render(collect_stats=True)[0], ref, "strip, instrumented kernels")
    for rank in range(2):
        part, _ = gpu.render(tile_rank=rank, tile_nranks=2, spp_per_pass=24 + 24 * rank)
        pref, _ = oracle.render(scene, tile_rank=rank, tile_nranks=2)
        assert_bitwise(part, pref, f"strip, shard {rank} of 2")


def test_pixelbounds_bitwise(binding, oracle, tmp_path):
    """"pixelbounds" of the path integrator (path.cpp:216-229, integrator.cpp:272; refused until round 6): pixels of the sample
    bounds outside the rectangle take no samples — no paths, no film weight — while tiles, sample indices and the film's sums keep
    their places. Film and counters against the oracle bit for bit: rectangles inside a tile, across tiles, ragged, one pixel, empty;
    the box film in one pass and in several, both kernel sets, two shards; a wide filter (samples near the rectangle's edge reach
    pixels outside it and pixels inside miss their outside neighbours'); Sobol'; and the strip of
    test_whole_number_film_positions_bitwise cut by a rectangle (whole-number film positions land in a neighbour that took no samples
    of its own, and the other way round: the exact finish's ordering). The IISPT entry points ignore it, as the reference's IISPT integrator does. Pins:
    tests/test_oracle_pins.py::test_pixelbounds_pins."""
    from test_oracle_pins import _killeroo_with
    X, Y, S = 112, 80, 5
    for k, (rect, flt, sampler) in enumerate(((( 21,  70,  9, 50), "", ""), ((16, 48, 16, 32), "", ""), ((17, 19, 33, 34), "", ""), ((0, 112, 40, 80), "", ""),
                                              ((90, 300, -4, 7), "", ""), ((200, 300, 0, 10), "", ""), ((21, 70, 9, 50), 'PixelFilter "gaussian"', ""),
                                              ((30, 33, 30, 33), 'PixelFilter "mitchell" "float xwidth" [3] "float ywidth" [1.5]', ""), ((21, 70, 9, 50), "", "sobol"))):
        line = 'Integrator "path" "integer pixelbounds" [%d %d %d %d]' % rect
        scene = binding.HostScene(path=_killeroo_with(tmp_path, line, X, Y, S, flt), **({"sampler": sampler} if sampler else {}))
        gpu = binding.GpuScene(scene)
        ref, ost = oracle.render(scene)
        film, st = gpu.render(collect_stats=True)
        assert_bitwise(film, ref, f"pixelbounds {rect} {flt} {sampler}: instrumented kernels")
        assert st["camera_rays"] == ost["camera_rays"] and st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
        assert_bitwise(gpu.render()[0], ref, f"pixelbounds {rect} {flt} {sampler}")
        assert_bitwise(gpu.render(spp_per_pass=2)[0], ref, f"pixelbounds {rect} {flt} {sampler}: three passes")
        if k == 0:
            for rank in range(2):
                part, _ = gpu.render(tile_rank=rank, tile_nranks=2)
                assert_bitwise(part, oracle.render(scene, tile_rank=rank, tile_nranks=2)[0], f"pixelbounds {rect}: shard {rank} of 2")
            # the IISPT integrator never looks at the parameter (its runners get film->GetSampleBounds(), iispt.cpp:395-409; the direct
            # integrator the film's bounds, iisptrenderrunner.cpp:608-613): the same direct pass as without it
            plain_scene = binding.HostScene(path=_killeroo_with(tmp_path, 'Integrator "path"', X, Y, S))
            assert np.array_equal(gpu.render_direct(2).view(np.uint64), oracle.iispt_direct(plain_scene, 2).view(np.uint64))
            # and through the C++ host: `iile_pbrt scene.pbrt` writes the same image
            import os
            import subprocess
            exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pbrt-v3-iile_amd", "lib", "iile_pbrt")
            out = tmp_path / "pb_cli.pfm"
            p = subprocess.run([exe, _killeroo_with(tmp_path, line, X, Y, S, flt), "--outfile", str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
            assert p.returncode == 0, p.stdout
            raw = out.read_bytes()
            head = b"PF\n%d %d\n-1.0\n" % (X, Y)
            assert raw.startswith(head)
            assert_bitwise(np.frombuffer(raw[len(head):], "<f4").reshape(Y, X, 3)[::-1], scene.film_to_rgb(ref), "pixelbounds through iile_pbrt")
        if rect[0] >= 200:
            assert (ref == 0).all() and ost["camera_rays"] == 0
    # whole-number film positions across the rectangle's edges (48 spp: dozens of them beyond x = 1024)
    for rect in ((1030, 1700, 3, 21), (1101, 1102, 0, 24), (0, 1900, 7, 8)):
        line = 'Integrator "path" "integer pixelbounds" [%d %d %d %d]' % rect
        scene = binding.HostScene(path=_killeroo_with(tmp_path, line, 1900, 24, 48))
        gpu = binding.GpuScene(scene)
        ref, _ = oracle.render(scene)
        assert_bitwise(gpu.render()[0], ref, f"strip with pixelbounds {rect}, one pass")
        assert_bitwise(gpu.render(spp_per_pass=10)[0], ref, f"strip with pixelbounds {rect}, five passes")
        assert_bitwise(gpu.render(collect_stats=True)[0], ref, f"strip with pixelbounds {rect}, instrumented kernels")
        if rect[0] == 1030:
            w = ref[..., 3]
            assert (w[:, :1030] > 0).sum() + (w[:, 1700:] > 0).sum() + (w[:3] > 0).sum() + (w[21:] > 0).sum() > 0   # a sample did land outside
            assert (w[3:21, 1030:1700] > 48).sum() > 10


def test_exact_finish_overflow_reaches_the_asynchronous_caller(binding, oracle):
    """The exact finish is sized from the frame; when it runs out of room anyway the film is wrong and the caller must hear of
    it. iile_render(film on the device, no statistics) only enqueues and returns IILE_OK, so the error has to come later:
    iile_render_status (waits for the stream) reports it once, and if nobody asks, the next iile_render does on entry. A render
    that waits (host film, or statistics) reports it itself. With the capacity sized from the frame again, the same strip is
    the oracle's bit for bit."""
    import torch
    torch.cuda.init()
    scene = binding.HostScene(xres=1900, yres=24, spp=48)   # dozens of whole-number film positions (test above)
    gpu = binding.GpuScene(scene)
    h, w = scene.film_shape
    film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    gpu.test_patch_capacity(8)   # room for 8 pixel hits: far too few
    gpu.render(film_device_ptr=film.data_ptr(), stream=stream, want_stats=False)   # returns IILE_OK: nothing has run yet
    with pytest.raises(RuntimeError, match="ran out of room"):
        gpu.render_status(stream)
    gpu.render_status(stream)   # reported once; the flag is cleared
    gpu.render(film_device_ptr=film.data_ptr(), stream=stream, want_stats=False)
    with pytest.raises(RuntimeError, match="previous asynchronous"):
        gpu.render(film_device_ptr=film.data_ptr(), stream=stream, want_stats=False)   # nobody asked: the next call does
    with pytest.raises(RuntimeError, match="ran out of room"):
        gpu.render()   # a render that waits reports its own overflow
    gpu.test_patch_capacity(0)
    ref, _ = oracle.render(scene)
    gpu.render(film_device_ptr=film.data_ptr(), stream=stream, want_stats=False)
    gpu.render_status(stream)
    assert_bitwise(film.cpu().numpy(), ref, "strip, capacity sized from the frame")


def test_device_film_is_finished_without_a_host_wait(binding, oracle):
    """The exact finish of the pixels reached by whole-number film positions runs on the device (kernels.hip "exact film
    finish"): iile_render with a device-resident film and no statistics only ENQUEUES — two renders and a reduction of their
    films queued back to back on one stream return to the host long before the GPU is done (rounds 1-3 waited for every pass
    there: VERDICT r03 "next" 5) — and the films are the oracle's bit for bit, as are those of the host-side finish kept behind
    IILE_DEBUG_HOST_FILM_FINISH, which does make the host wait."""
    import os
    import time
    import torch
    torch.cuda.init()
    scene = binding.HostScene(xres=1900, yres=64, spp=48)   # past x = 1024: rounding makes whole-number positions (above)
    gpu = binding.GpuScene(scene)
    ref, _ = oracle.render(scene)
    h, w = scene.film_shape
    a = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    b = torch.zeros_like(a)
    stream = torch.cuda.current_stream().cuda_stream

    def enqueue_two():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        t0 = time.perf_counter()
        gpu.render(film_device_ptr=a.data_ptr(), stream=stream, want_stats=False)
        gpu.render(film_device_ptr=b.data_ptr(), stream=stream, want_stats=False, spp_per_pass=16)   # three passes
        total = a + b                                                                             # "the film reduce"
        host_ms = (time.perf_counter() - t0) * 1e3
        e1.record()
        torch.cuda.synchronize()
        return host_ms, e0.elapsed_time(e1), total

    enqueue_two()  # workspace allocation, tile tables
    host_ms, gpu_ms, total = enqueue_two()
    assert_bitwise(a.cpu().numpy(), ref, "device finish, one pass")
    assert_bitwise(b.cpu().numpy(), ref, "device finish, three passes")
    assert torch.equal(total, a + b)
    assert gpu_ms > 3.0 and host_ms < 0.5 * gpu_ms, (host_ms, gpu_ms)   # the host was back while the GPU still worked
    os.environ["IILE_DEBUG_HOST_FILM_FINISH"] = "1"
    try:
        host_ms2, gpu_ms2, _ = enqueue_two()
        assert_bitwise(a.cpu().numpy(), ref, "host finish, one pass")
        assert_bitwise(b.cpu().numpy(), ref, "host finish, three passes")
        assert host_ms2 > 0.5 * gpu_ms2, (host_ms2, gpu_ms2)              # that version waits for every pass
    finally:
        del os.environ["IILE_DEBUG_HOST_FILM_FINISH"]


def test_two_stream_schedule_is_the_one_stream_film(gpu_small, scene_small, oracle):
    """The NEE kernels of a bounce run on a second stream beside the next bounce's k_extend / k_shade (doubled NEE
    records); time_kernels = 2 puts every kernel on the caller's stream. Same film, also across several passes, and the
    per-kernel times of the overlapped schedule sum to at least the serial ones' order of magnitude."""
    ref, _ = oracle.render(scene_small)
    two, st2 = gpu_small.render(time_kernels=1)
    one, st1 = gpu_small.render(time_kernels=2)
    assert_bitwise(two, ref, "two streams")
    assert_bitwise(one, ref, "one stream")
    assert st1["ms_shade"] > 0 and st2["ms_shade"] > 0 and st1["ms_shadow"] > 0 and st2["ms_shadow"] > 0
    chunked, st = gpu_small.render(spp_per_pass=1)
    assert st["n_passes"] > 1
    assert_bitwise(chunked, ref, "two streams, several passes")
    for _ in range(3):  # the schedule is not deterministic, the film is
        assert_bitwise(gpu_small.render()[0], ref, "repeat")


def test_edge_sizes_depths_and_crops_bitwise(binding, oracle):
    """The small and ragged ends of the tile loop (integrator.cpp:235-330): a one-pixel film, a film narrower than a tile
    with ragged tiles, a rank with no tile at all, maxdepth 0 (camera ray + emitted light only) and 1, a crop window whose
    sample bounds start inside a tile grid of their own (film.cpp:47-61) — film and counters, both kernel sets."""
    import os
    src = open(binding.DEFAULT_SCENE).read()

    def check(scene, what, ranks=()):
        gpu = binding.GpuScene(scene)
        ref, ost = oracle.render(scene)
        film, st = gpu.render(collect_stats=True)
        assert_bitwise(film, ref, what + ": film, instrumented")
        assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"] and st["camera_rays"] == ost["camera_rays"]
        assert_bitwise(gpu.render()[0], ref, what + ": film")
        for rank, n in ranks:
            part, pst = gpu.render(tile_rank=rank, tile_nranks=n)
            pref, _ = oracle.render(scene, tile_rank=rank, tile_nranks=n)
            assert_bitwise(part, pref, f"{what}: shard {rank}/{n}")
        return gpu

    check(binding.HostScene(xres=1, yres=1, spp=1), "1 x 1, 1 spp", ranks=((0, 8), (3, 8), (7, 8)))
    check(binding.HostScene(xres=1, yres=1, spp=7), "1 x 1, 7 spp")
    check(binding.HostScene(xres=37, yres=5, spp=3), "37 x 5, 3 spp", ranks=((0, 2), (1, 2), (4, 5)))
    for depth in (0, 1):
        path = os.path.join(os.path.dirname(binding.DEFAULT_SCENE), f"_killeroo_gpu_depth{depth}.pbrt")
        text = src.replace('Integrator "path"', 'Integrator "path" "integer maxdepth" [%d]' % depth, 1)
        assert text != src
        open(path, "w").write(text)
        try:
            scene = binding.HostScene(path=path, xres=96, yres=64, spp=2)
        finally:
            os.remove(path)
        assert scene.info["max_depth"] == depth
        check(scene, f"maxdepth {depth}")
    path = os.path.join(os.path.dirname(binding.DEFAULT_SCENE), "_killeroo_gpu_crop.pbrt")
    text = src.replace('"integer yresolution"', '"float cropwindow" [0.21 0.83 0.3 0.66] "integer yresolution"', 1)
    assert text != src
    open(path, "w").write(text)
    try:
        scene = binding.HostScene(path=path, xres=200, yres=150, spp=3)
    finally:
        os.remove(path)
    f = scene.film
    assert 0 < f.crop_x0 < f.crop_x1 < 200 and 0 < f.crop_y0 < f.crop_y1 < 150 and f.crop_x0 % 16 != 0
    check(scene, "crop window", ranks=((0, 3), (2, 3)))


def test_cpp_cli_writes_the_exr_the_scene_names(scene_small, gpu_small, tmp_path):
    """Without --outfile iile_pbrt writes the scene file's Film "filename" (killeroo-simple.exr), as pbrt does: half-float RGB
    of the same film the C ABI renders."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "pbrt-v3-iile_amd", "lib", "iile_pbrt")
    p = subprocess.run([exe, os.path.join(repo, "scenes", "killeroo-simple.pbrt"), "--xres", "160", "--yres", "120", "--spp", "4"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout
    out = tmp_path / "killeroo-simple.exr"
    assert out.exists(), p.stdout
    img = __import__("importlib").import_module("pbrt-v3-iile_amd.binding").read_image(str(out))
    film, _ = gpu_small.render()
    want = scene_small.film_to_rgb(film).astype(np.float16).astype(np.float32)
    assert_bitwise(img, want, "CLI image through OpenEXR")


def test_cpp_cli_takes_pbrts_own_options(tmp_path):
    """iile_pbrt accepts pbrt's command line (src/main/pbrt.cpp:106-186): --quick renders a quarter of the resolution at one
    sample per pixel, --outfile= / --nthreads / --quiet / the logging flags are taken as pbrt spells them."""
    import importlib
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "pbrt-v3-iile_amd", "lib", "iile_pbrt")
    out = tmp_path / "q.pfm"
    p = subprocess.run([exe, "--quick", "--quiet", "--nthreads", "8", "--nthreads=4", "--logtostderr", "--v=1", "--outfile=" + str(out),
                        os.path.join(repo, "scenes", "killeroo-simple.pbrt")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "rendered" not in p.stdout, p.stdout
    b = importlib.import_module("pbrt-v3-iile_amd.binding")
    img = b.read_image(str(out))
    scene = b.HostScene(quick=True)
    assert img.shape == (scene.info["yres"], scene.info["xres"], 3) == (175, 175, 3) and scene.info["spp"] == 1
    film, _ = b.GpuScene(scene).render()
    assert_bitwise(img, scene.film_to_rgb(film), "--quick image")
