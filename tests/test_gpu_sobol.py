"""SobolSampler on the device against the oracle, through the C ABI: sample indices, sample values of
every dimension a path can consume, the film of BASELINE config 0's frame rendered with it, tile sharding and passes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (a == b)).all()


@pytest.fixture(scope="module")
def scene_sobol(binding):
    return binding.HostScene(xres=400, yres=400, spp=8, sampler="sobol")


@pytest.fixture(scope="module")
def gpu_sobol(binding, scene_sobol):
    return binding.GpuScene(scene_sobol)


def test_sobol_samples_match_oracle(gpu_sobol, scene_sobol, oracle):
    pix = [(0, 0), (5, 7), (127, 127), (128, 130), (399, 399), (255, 1), (17, 300)]
    ks = [0, 1, 7]
    px = np.array([p[0] for p in pix for _ in ks], np.int32)
    py = np.array([p[1] for p in pix for _ in ks], np.int32)
    k = np.array([kk for _ in pix for kk in ks], np.int32)
    ndims = 48
    dev, idx = gpu_sobol.halton_samples(px, py, k, 0, ndims)
    for i in range(len(px)):
        ref_idx = oracle.sample_index(scene_sobol, px[i], py[i], k[i])
        assert int(idx[i]) == ref_idx
        ref = np.array([oracle.sample_dimension(scene_sobol, ref_idx, d, px[i], py[i]) for d in range(ndims)], np.float32)
        assert _bits_equal(dev[i], ref), (pix[i // len(ks)], k[i])


def test_sobol_li_per_sample(gpu_sobol, scene_sobol, oracle):
    rng = np.random.default_rng(11)
    n = 2048
    px = rng.integers(0, 400, n).astype(np.int32)
    py = rng.integers(0, 400, n).astype(np.int32)
    k = rng.integers(0, 8, n).astype(np.int32)
    L, nr = gpu_sobol.li_samples(px, py, k)
    rL, rnr = oracle.li(scene_sobol, px, py, k)
    assert np.array_equal(nr, rnr), "per-sample ray counts differ"
    assert _bits_equal(L, rL)


def test_sobol_film_c1_bitwise(gpu_sobol, scene_sobol, oracle):
    """killeroo-simple 400x400 x 8 spp with the Sobol' sampler: film and every counter equal to the oracle's, with the
    instrumented and the plain kernels, in one pass and in several, whole and sharded."""
    ref, ost = oracle.render(scene_sobol)
    film, st = gpu_sobol.render(collect_stats=True)
    assert _bits_equal(film, ref)
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["nodes_closest"] == ost["nodes_closest"] and st["tri_tests"] == ost["tri_tests"]
    plain, _ = gpu_sobol.render()
    assert _bits_equal(plain, ref)
    chunked, cst = gpu_sobol.render(spp_per_pass=2)
    assert cst["n_passes"] >= 4 and _bits_equal(chunked, ref)
    acc = np.zeros_like(ref)
    for r in range(3):
        part, _ = gpu_sobol.render(tile_rank=r, tile_nranks=3)
        pref, _ = oracle.render(scene_sobol, tile_rank=r, tile_nranks=3)
        assert _bits_equal(part, pref)
        acc += part
    assert np.array_equal(acc[..., 3], ref[..., 3])


def test_sobol_in_a_textured_room(binding, oracle, tmp_path):
    """The EXT / TEX builds of the shade kernel with the Sobol' sampler (specular materials, several lights, image
    textures: every sampler call site of the wider feature set)."""
    import boxroom
    path = tmp_path / "room.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, light="envmap", materials="mixed", textures=str(tmp_path)))
    scene = binding.HostScene(path=str(path), sampler="sobol")
    gpu = binding.GpuScene(scene)
    ref, _ = oracle.render(scene)
    film, _ = gpu.render()
    assert _bits_equal(film, ref)
    counted, _ = gpu.render(collect_stats=True)
    assert _bits_equal(counted, ref)
