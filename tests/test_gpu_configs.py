"""BASELINE.json's configurations at their real sizes, on one GPU (marked gpu).

config 2  killeroo-simple 1920x1080 x 64 spp, one GPU          -> test_config2_full_frame_vs_oracle (the whole frame),
                                                                  test_config2_1080p_frame_bitwise
config 3  1920x1080 x 1024 spp, film tiles sharded over 8 GPUs -> test_config3_1080p_1024spp_eight_shards
                                                                  (the eight ranks' renders run back to back on the one GPU)
config 4  Sponza-class deep BVH at 1080p                        -> test_config4_full_size_room_1080p
          (Sponza does not ship with the reference: the 287 k-triangle synthetic room of tests/boxroom.py stands in)
The film merge itself (RCCL) can only be exercised at world size 1 on a 1-GPU box: test_dist_world1_film_reduce.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torch():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    return torch


def _bits_equal(a, b):
    same = (a.view(np.uint32) == b.view(np.uint32)) | (a == b)
    return same


def test_config2_1080p_frame_bitwise(binding, oracle):
    """The bench frame's geometry at 8 of its 64 samples against the oracle (film + every counter), and the full
    64 spp in one pass against eight 8-spp passes."""
    scene8 = binding.HostScene(xres=1920, yres=1080, spp=8)
    gpu8 = binding.GpuScene(scene8)
    film, st = gpu8.render(collect_stats=True)
    ref, ost = oracle.render(scene8)
    assert _bits_equal(film, ref).all()
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["nodes_closest"] == ost["nodes_closest"] and st["nodes_any"] == ost["nodes_any"] and st["tri_tests"] == ost["tri_tests"]
    plain, _ = gpu8.render()
    assert _bits_equal(plain, ref).all()
    del gpu8
    scene64 = binding.HostScene(xres=1920, yres=1080, spp=64)
    gpu64 = binding.GpuScene(scene64)
    one, st1 = gpu64.render()
    many, st8 = gpu64.render(spp_per_pass=8)
    assert st1["n_passes"] == 1 and st8["n_passes"] >= 8
    assert _bits_equal(one, many).all()


def test_config2_full_frame_vs_oracle(binding, oracle):
    """The headline frame itself under the oracle: 1920x1080 x 64 spp (132.7 M camera samples, 765 M reference rays),
    rendered by the kernels bench.py times (plain build: four-wide BVH steps, MIS-ray culling, dense MIS queue, two
    streams, camera rays made in k_extend, the last bounce left out) AND by the instrumented build, against the CPU
    oracle on all host cores — film {X, Y, Z, weight} bit for bit (1e-4 relative is the north star's tolerance; the test
    asks for equality), every traversal counter, the path-length histogram. 8 spp does not stand in for 64: whole-number
    film positions only appear past sample 8 (round 1's bug). SamplerIntegrator::Render, integrator.cpp:227-339."""
    scene = binding.HostScene(xres=1920, yres=1080, spp=64)
    gpu = binding.GpuScene(scene)
    plain, pst = gpu.render()
    assert pst["n_passes"] == 1
    counted, cst = gpu.render(collect_stats=True)
    ref, ost = oracle.render(scene)
    assert ost["camera_rays"] == 1920 * 1080 * 64
    for name, film in (("plain kernels", plain), ("instrumented kernels", counted)):
        same = _bits_equal(film, ref)
        assert same.all(), f"{name}: {int((~same).any(-1).sum())} pixels differ from the oracle"
    assert cst["camera_rays"] == ost["camera_rays"]
    assert cst["closest_rays"] == ost["regular_rays"] and cst["shadow_rays"] == ost["shadow_rays"]
    assert cst["nodes_closest"] == ost["nodes_closest"] and cst["nodes_any"] == ost["nodes_any"]
    assert cst["tri_tests"] == ost["tri_tests"] and cst["tri_hits"] == ost["tri_hits"]
    assert list(cst["path_length"]) == list(ost["path_length"])
    assert cst["nee_evals"] == ost["nee_evals"] and cst["zero_radiance"] == ost["zero_radiance"]
    # what the plain build does not trace is exactly what it may leave out: fewer rays, the same film
    assert pst["ext_rays_traced"] + pst["mis_rays_traced"] < cst["closest_rays"]


def test_config3_1080p_1024spp_eight_shards(binding, oracle):
    """BASELINE config 3 without the eight GPUs: ranks 0..7 of 8 render their tiles of the 1920x1080 x 1024 spp frame
    one after the other; their films, summed as the RCCL reduction sums them, must be the single-call 1024-spp film:
    weights exactly, values bitwise wherever a pixel cannot receive from another rank's tile (tile interiors), to 1e-6
    relative on tile borders (whole-number film positions splat across tiles; more than two float summands there may be
    added in another order). Each shard is also bitwise the oracle's shard at 8 spp."""
    torch = _torch()
    scene = binding.HostScene(xres=1920, yres=1080, spp=1024)
    gpu = binding.GpuScene(scene)
    h, w = scene.film_shape
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    _, st = gpu.render(film_device_ptr=full.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert st["n_paths"] == 120 * 68 * 256 * 1024  # path slots of the 120 x 68 tiles (the last tile row is half outside the frame)
    acc = torch.zeros_like(full)
    part = torch.zeros_like(full)
    owners = torch.full((h, w), -1, dtype=torch.int32, device="cuda")
    n_paths = 0
    for r in range(8):
        _, ps = gpu.render(tile_rank=r, tile_nranks=8, film_device_ptr=part.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        n_paths += ps["n_paths"]
        acc += part
        own = part[..., 3] >= 1024  # a rank's own pixels carry all 1024 weights (a splat adds a few)
        assert int((owners[own] >= 0).sum()) == 0, "two ranks own a pixel"
        owners[own] = r
    assert n_paths == 120 * 68 * 256 * 1024
    assert int((owners < 0).sum()) == 0
    # the ownership the films show is iile_tile_owner: diagonal interleave of the 16x16 tiles
    ty, tx = torch.meshgrid(torch.arange(h, device="cuda") // 16, torch.arange(w, device="cuda") // 16, indexing="ij")
    assert torch.equal(owners, ((tx + ty) % 8).to(torch.int32))
    assert torch.equal(acc[..., 3], full[..., 3])
    yy, xx = torch.meshgrid(torch.arange(h, device="cuda") % 16, torch.arange(w, device="cuda") % 16, indexing="ij")
    interior = (yy > 0) & (yy < 15) & (xx > 0) & (xx < 15)
    assert torch.equal(acc[interior].view(torch.int32), full[interior].view(torch.int32))
    assert torch.allclose(acc, full, rtol=1e-6, atol=0)
    del gpu, acc, part, full
    # every shard against the oracle (8 spp keeps the CPU side to seconds)
    scene8 = binding.HostScene(xres=1920, yres=1080, spp=8)
    gpu8 = binding.GpuScene(scene8)
    for r in range(8):
        shard, _ = gpu8.render(tile_rank=r, tile_nranks=8)
        ref, _ = oracle.render(scene8, tile_rank=r, tile_nranks=8)
        assert _bits_equal(shard, ref).all(), f"shard {r}/8 differs from the oracle"


def test_config4_full_size_room_1080p(binding, oracle, tmp_path):
    """The full-size stand-in for the Sponza configuration: the 287 k-triangle closed room (the scene of
    `bench.py --workload boxroom`) at 1920x1080, one sample per pixel, against the oracle — film and counters with the
    instrumented kernels (binary BVH steps), film with the plain ones (four-wide steps)."""
    import boxroom
    path = tmp_path / "room.pbrt"
    path.write_text(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64, xres=1920, yres=1080, spp=1))
    scene = binding.HostScene(path=str(path))
    assert scene.info["n_triangles"] > 250000
    gpu = binding.GpuScene(scene)
    film, st = gpu.render(collect_stats=True)
    ref, ost = oracle.render(scene)
    assert _bits_equal(film, ref).all()
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["nodes_closest"] == ost["nodes_closest"] and st["nodes_any"] == ost["nodes_any"]
    assert st["tri_tests"] == ost["tri_tests"] and st["path_length"] == ost["path_length"]
    assert st["nodes_closest"] / st["closest_rays"] > 60  # deep tree, every path runs to maxdepth
    plain, _ = gpu.render()
    assert _bits_equal(plain, ref).all()


def test_config4_room_past_sample_8(binding, oracle, tmp_path):
    """The same deep tree at 960x540 and 16 samples per pixel: past sample 8 the Halton samples of a pixel meet
    whole-number film positions (round 1's film bug only showed there) and the adaptive refill of the persistent traversal
    waves has run through many queue generations. Film and every counter with the instrumented kernels, film with the plain
    ones, and the plain kernels again in passes of 5 samples' worth of paths — all against the oracle, bit for bit."""
    import boxroom
    path = tmp_path / "room16.pbrt"
    path.write_text(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64, xres=960, yres=540, spp=16))
    scene = binding.HostScene(path=str(path))
    assert scene.info["n_triangles"] > 250000 and scene.info["spp"] == 16
    gpu = binding.GpuScene(scene)
    ref, ost = oracle.render(scene)
    film, st = gpu.render(collect_stats=True)
    assert _bits_equal(film, ref).all()
    assert st["camera_rays"] == 960 * 540 * 16 == ost["camera_rays"]
    assert st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert st["nodes_closest"] == ost["nodes_closest"] and st["nodes_any"] == ost["nodes_any"]
    assert st["tri_tests"] == ost["tri_tests"] and st["path_length"] == ost["path_length"]
    plain, pst = gpu.render()
    assert pst["n_passes"] == 1 and _bits_equal(plain, ref).all()
    passes, pst = gpu.render(spp_per_pass=5)
    assert pst["n_passes"] >= 3 and _bits_equal(passes, ref).all()


def test_dist_world1_film_reduce(binding, gpu_small, scene_small):
    """libiile_dist.so on hardware, as far as one GPU allows: a one-rank communicator, the in-place film reduction on
    the stream the render ran on, the host-side totals."""
    torch = _torch()
    comm = binding.Dist(binding.Dist.unique_id(), 0, 1)
    h, w = scene_small.film_shape
    film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    gpu_small.render(film_device_ptr=film.data_ptr(), stream=stream)
    comm.film_reduce(film.data_ptr(), h * w, 0, stream)
    comm.barrier(stream)
    torch.cuda.synchronize()
    host, _ = gpu_small.render()
    assert np.array_equal(film.cpu().numpy().view(np.uint32), host.view(np.uint32))
    assert comm.sum_u64([3, 2 ** 40]).tolist() == [3, 2 ** 40]
    assert comm.max_f64([1.5, -2.0]).tolist() == [1.5, -2.0]
    comm.close()
