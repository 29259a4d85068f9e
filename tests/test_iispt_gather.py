"""The IISPT render runner's gather (second half; src/integrators/iisptrenderrunner.cpp:216-596).

CPU: properties of the oracle's restatement (the reference has no test for the runner and no weights for the network, so
parity of these functions rests on the restatement plus these properties). GPU: the device kernels against the oracle,
bit for bit, on seeded synthetic "predictions" (the gather needs no network weights to be tested)."""
import numpy as np
import pytest


def _task(binding, w, h, ts, counter=0, seed=99):
    return binding.IisptTask(0, 0, w, h, ts, counter, seed)


def _bits_equal(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (a == b)).all()


def test_hemi_grid_of_a_task(binding):
    """Hemi points sit every `tilesize` pixels and on the last row / column (iisptrenderrunner.cpp:387-412)."""
    for (a1, ts, want) in ((96, 10, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95]), (81, 10, [0, 10, 20, 30, 40, 50, 60, 70, 80]),
                           (5, 10, [0, 4]), (1, 3, [0])):
        t = binding.IisptTask(0, 0, a1, a1, ts, 0, 0)
        nx, ny = t.grid()
        assert nx == ny == len(want)


def test_oracle_gather_properties(binding, oracle):
    scene = binding.HostScene(xres=96, yres=80, spp=4)
    task = _task(binding, 96, 80, 10)
    valid, pos, dr = oracle.iispt_hemi_points(scene, task)
    assert valid.shape == (9, 11) and valid.sum() > 0
    # a valid hemi point's aux ray leaves along the unit normal
    n = np.linalg.norm(dr[valid == 1], axis=1)
    assert np.allclose(n, 1, atol=1e-5) and (dr[valid == 0] == 0).all()
    rng = np.random.default_rng(5)
    nn = rng.uniform(0.1, 2.0, valid.shape + (32, 32, 3)).astype(np.float32)
    out = oracle.iispt_gather(scene, task, valid, pos, dr, nn)
    assert np.isfinite(out).all() and (out[..., :3] >= 0).all()
    w = out[..., 3]
    assert set(np.unique(w)) <= {0.0, 0.5} and (w == 0.5).sum() > 1000
    assert (out[w == 0][:, :3] == 0).all()
    # radiance is linear in the predicted hemispheres: a power-of-two factor carries through every float product exactly
    out2 = oracle.iispt_gather(scene, task, valid, pos, dr, nn * np.float32(4))
    assert np.array_equal(out2[..., :3], out[..., :3] * np.float32(4))
    # dark hemispheres give black pixels that are still recorded; hemi points without a camera still take their share of
    # the 16 draws per neighbour (samples_taken) but add nothing
    zero = oracle.iispt_gather(scene, task, valid, pos, dr, nn * 0)
    assert (zero[..., :3] == 0).all() and np.array_equal(zero[..., 3], w)
    none = oracle.iispt_gather(scene, task, np.zeros_like(valid), pos, dr, nn)
    assert (none[..., :3] == 0).all() and np.array_equal(none[..., 3], w)
    # another RNG stream: a different estimate of the same quantity
    other = oracle.iispt_gather(scene, _task(binding, 96, 80, 10, seed=7), valid, pos, dr, nn)
    assert not np.array_equal(other, out)
    m0, m1 = out[..., :3][w > 0].mean(), other[..., :3][w > 0].mean()
    assert abs(m0 - m1) / m0 < 0.05


@pytest.mark.gpu
def test_gather_on_device_matches_oracle(binding, oracle):
    scene = binding.HostScene(xres=96, yres=80, spp=4)
    gpu = binding.GpuScene(scene)
    for task in (_task(binding, 96, 80, 10), binding.IisptTask(16, 8, 90, 71, 7, 500, 3)):
        valid, pos, dr = gpu.iispt_hemi_points(task)
        rv, rp, rd = oracle.iispt_hemi_points(scene, task)
        assert np.array_equal(valid, rv) and _bits_equal(pos, rp) and _bits_equal(dr, rd)
        rng = np.random.default_rng(17)
        nn = rng.uniform(0.0, 3.0, valid.shape + (32, 32, 3)).astype(np.float32)
        nn[rng.uniform(size=nn.shape[:4]) < 0.1] = 0  # black texels: the is_black branches
        out = gpu.iispt_gather(task, valid, pos, dr, nn)
        ref = oracle.iispt_gather(scene, task, valid, pos, dr, nn)
        assert _bits_equal(out, ref)
        assert (out[..., 3] == 0.5).sum() > 500


@pytest.mark.gpu
def test_batched_calls_equal_the_single_task_calls(binding, oracle):
    """iile_iispt_hemi_points_batch / iile_iispt_gather_batch: several tasks (different rectangles, tile sizes, sampler counters
    and seeds; one of a single pixel) from one set of launches — the single-task results, concatenated, bit for bit; and the
    oracle's for one of them."""
    scene = binding.HostScene(xres=96, yres=80, spp=4)
    gpu = binding.GpuScene(scene)
    tasks = [_task(binding, 96, 80, 10), binding.IisptTask(16, 8, 90, 71, 7, 500, 3), binding.IisptTask(40, 40, 41, 41, 5, 9000, 77),
             binding.IisptTask(0, 50, 96, 80, 30, 12000, 5), binding.IisptTask(60, 0, 96, 33, 4, 20000, 11)]
    singles = [gpu.iispt_hemi_points(t) for t in tasks]
    valid, pos, dr = gpu.iispt_hemi_points_batch(tasks)
    assert np.array_equal(valid, np.concatenate([s[0].reshape(-1) for s in singles]))
    assert _bits_equal(pos, np.concatenate([s[1].reshape(-1, 3) for s in singles]))
    assert _bits_equal(dr, np.concatenate([s[2].reshape(-1, 3) for s in singles]))
    rng = np.random.default_rng(31)
    nn = rng.uniform(0.0, 3.0, (len(valid), 32, 32, 3)).astype(np.float32)
    nn[rng.uniform(size=nn.shape[:3]) < 0.1] = 0
    out = gpu.iispt_gather_batch(tasks, valid, pos, dr, nn)
    first_h = first_p = 0
    for t, (v, p, d) in zip(tasks, singles):
        nh, npix = v.size, (t.x1 - t.x0) * (t.y1 - t.y0)
        one = gpu.iispt_gather(t, v, p, d, nn[first_h:first_h + nh].reshape(v.shape + (32, 32, 3)))
        assert _bits_equal(out[first_p:first_p + npix], one.reshape(-1, 4))
        if t is tasks[1]:
            assert _bits_equal(one, oracle.iispt_gather(scene, t, v, p, d, nn[first_h:first_h + nh].reshape(v.shape + (32, 32, 3))))
        first_h += nh
        first_p += npix
    assert first_p == len(out) and (out[:, 3] == 0.5).sum() > 2000


@pytest.mark.gpu
def test_gather_in_a_textured_room_with_specular_chains(binding, oracle, tmp_path):
    """find_intersection's specular chain (mirror / glass blobs), textured and bump-mapped first hits, plastic and uber BSDFs
    under sample_hemisphere; the predictions stay in HBM (device pointers in and out)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    import boxroom
    path = tmp_path / "room.pbrt"
    path.write_text(boxroom.boxroom_pbrt(xres=96, yres=64, spp=4, light="envmap", materials="mixed", textures=str(tmp_path)))
    scene = binding.HostScene(path=str(path))
    gpu = binding.GpuScene(scene)
    task = _task(binding, 96, 64, 12, counter=40, seed=2024)
    valid, pos, dr = gpu.iispt_hemi_points(task)
    rv, rp, rd = oracle.iispt_hemi_points(scene, task)
    assert np.array_equal(valid, rv) and _bits_equal(pos, rp) and _bits_equal(dr, rd)
    rng = np.random.default_rng(23)
    nn = rng.uniform(0.0, 1.5, valid.shape + (32, 32, 3)).astype(np.float32)
    nn_t = torch.from_numpy(nn).cuda()
    out_t = torch.zeros((64, 96, 4), dtype=torch.float32, device="cuda")
    gpu.iispt_gather(task, valid, pos, dr, nn_device_ptr=nn_t.data_ptr(), out_device_ptr=out_t.data_ptr())
    torch.cuda.synchronize()
    ref = oracle.iispt_gather(scene, task, valid, pos, dr, nn)
    assert _bits_equal(out_t.cpu().numpy(), ref)


@pytest.mark.gpu
def test_iispt_frame_end_to_end(binding):
    """BASELINE config 5's data flow on a small frame: schedule -> hemi points -> probe pass -> network (random weights) ->
    gather -> film monitor, everything resident in HBM. With an untrained network only structure can be asserted: every
    pixel whose camera ray finds a scattering surface gets exactly one sample of weight 0.5 per sweep, values are finite
    and non-negative, and a second sweep (smaller radius, other camera samples and random streams) adds as much again."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    scene = binding.HostScene(xres=96, yres=80, spp=1)
    gpu = binding.GpuScene(scene)
    torch.manual_seed(0)
    import iispt_torch_reference as ref_mod
    pipe = nn_mod.IisptPipeline(gpu, net=ref_mod.IISPTNet())
    frame = frame_mod.IisptFrame(binding, gpu, pipe)
    tasks = list(frame_mod.schedule((0, 0, 96, 80), 100, radius_start=4.0))
    sweep = [t for t in tasks if t[4] == 4]
    assert len(sweep) == 3 * 2 and sweep[0] == (0, 0, 40, 40, 4) and sweep[-1] == (80, 40, 96, 80, 4)
    for t in sweep:
        frame.run_task(*t)
    w1 = frame.film[..., 3].clone()
    assert set(torch.unique(w1).tolist()) <= {0.0, 0.5} and float((w1 == 0.5).float().mean()) > 0.9
    img = frame.indirect_image()
    assert bool(torch.isfinite(img).all()) and float(img.min()) >= 0
    assert frame.stats["pixels"] == 96 * 80 and frame.stats["probes"] > 0
    n3 = [t for t in tasks if t[4] == 3]
    for t in n3[: (-(-96 // 30)) * (-(-80 // 30))]:
        frame.run_task(*t)
    w2 = frame.film[..., 3]
    assert set(torch.unique(w2).tolist()) <= {0.0, 0.5, 1.0}
    assert abs(float(w2.mean()) / float(w1.mean()) - 2.0) < 0.02  # (a silhouette pixel may find a surface in one sweep only)
    # task-major staging (one probe pass + one network call per group of tasks) is the same frame: the same samples are
    # recorded (weights exactly equal) and the values agree — the network's kernels do not depend on the batch (bit for bit:
    # tests/test_iispt_nn.py); what is left is the film monitor adding a pixel's two sweeps in one order or the other
    batched = frame_mod.IisptFrame(binding, gpu, pipe)
    batched.run_batched(len(sweep) + (-(-96 // 30)) * (-(-80 // 30)), radius_start=4.0, max_probes=150)
    assert torch.equal(batched.film[..., 3], w2) and batched.stats == frame.stats
    scale = float(frame.film[..., :3].abs().max())
    assert float((batched.film[..., :3] - frame.film[..., :3]).abs().max()) <= 1e-6 * scale
    # the direct pass and the merge (iispt.cpp:405-446): the direct monitor is the oracle's bit for bit, also when it is filled
    # in two calls; the frame's image is merge_into + to_intensity_film of the two monitors as the oracle computes it
    import oracle_binding
    orc = oracle_binding.Oracle()
    frame.run_direct(2)
    frame.run_direct(1)
    torch.cuda.synchronize()
    ref_direct = orc.iispt_direct(scene, 3)
    assert np.array_equal(frame.film_direct.cpu().numpy().view(np.uint64), ref_direct.view(np.uint64))
    merged = orc.iispt_merge(ref_direct, frame.film.cpu().numpy())
    assert np.array_equal(frame.image().cpu().numpy().view(np.uint32), merged.view(np.uint32))
    assert float(frame.image().mean()) > float(frame.indirect_image().mean())
