"""SURVEY.md §8 f3, the IISPT integrator's direct pass and the final merge: DirectProgressiveIntegrator::Li / RenderOnePass
(src/integrators/directprogressiveintegrator.cpp:22-150) driven by IisptRenderRunner::run_direct
(src/integrators/iisptrenderrunner.cpp:601-633), IisptFilmMonitor::merge_into (src/integrators/iisptfilmmonitor.cpp:231-275).
No reference test, output or trained network exists for the fork's own integrator: the oracle's restatement is held to
analytic values here (CPU); the device pass to the oracle bit for bit in the GPU tests below."""
import os

import numpy as np
import pytest

import oracle_binding as ob

SCENES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes")


def _mean_rgb(film):
    return (film[..., :3] / film[..., 3:4]).mean(axis=(0, 1))


def test_direct_pass_analytic_values(binding, oracle):
    """Inside the reference's analytic furnace scenes (src/tests/analytic_scenes.cpp) the direct pass has closed forms:
    point light I = pi at the centre of a Kd = 0.5 unit sphere: Kd / pi * I / r^2 = 0.5 exactly (one light sample, no
    visibility); Kd = 0.5, Le = 0.5 emitter: Le + Kd * Le = 0.75 (emitted + one bounce of direct light, Monte Carlo);
    UberMaterial Kd = 0.25, Kr = 0.5 with eta 1 under a point light 3 pi: 0.75 (FresnelDielectric(1, 1) = 0 kills the
    specular recursion); the emitting mirror ball: 0.5 * (1 + .5 + .25 + .125 + .0625) = 0.96875 — the recursion's depth
    limit `depth + 1 < maxDepth`, deterministic."""
    for name, want, tol in (("furnace_point.pbrt", 0.5, 1e-5), ("furnace_area.pbrt", 0.75, 0.02), ("furnace_uber.pbrt", 0.75, 1e-5),
                            ("furnace_mirror_emitter.pbrt", 0.96875, 1e-5)):
        scene = binding.HostScene(path=os.path.join(SCENES, name))
        film = oracle.iispt_direct(scene, 16, trig_mode=ob.TRIG_LIBM)
        assert (film[..., 3] == 16.0).all()
        m = _mean_rgb(film)
        assert np.allclose(m, want, rtol=0, atol=tol), (name, m)


def test_direct_pass_is_a_sum_of_independent_passes(binding, oracle):
    """add_n_samples adds pass after pass in double precision: passes [0, 3) equal pass 0 + pass 1 + pass 2 added in that
    order, bit for bit; different passes draw different samples; thread count does not matter."""
    scene = binding.HostScene(xres=48, yres=36, spp=1)
    all3 = oracle.iispt_direct(scene, 3, threads=3)
    acc = np.zeros_like(all3)
    singles = []
    for p in range(3):
        one = oracle.iispt_direct(scene, 1, first_pass=p, threads=1)
        singles.append(one)
        acc += one
    assert np.array_equal(acc.view(np.uint64), all3.view(np.uint64))
    assert not np.array_equal(singles[0], singles[1])
    # the image is killeroo-simple under direct light only: darker than the path-traced one, not black
    path, _ = oracle.render(scene)
    assert 0.2 < _mean_rgb(all3).mean() < scene.film_to_rgb(path).mean() * 1.05


def test_merge_normalises_both_monitors(oracle):
    """merge_into: each monitor's sums over its own weight, then added; a pixel one monitor never saw counts as 0."""
    rng = np.random.default_rng(5)
    d = rng.random((4, 5, 4))
    i = rng.random((4, 5, 4))
    d[..., 3] = 16.0
    i[..., 3] = rng.integers(1, 40, (4, 5)).astype(np.float64)
    i[0, 0] = 0.0  # never written by the indirect pass
    out = oracle.iispt_merge(d, i)
    want = d[..., :3] / d[..., 3:4] + np.where(i[..., 3:4] > 0, i[..., :3] / np.where(i[..., 3:4] > 0, i[..., 3:4], 1.0), i[..., :3])
    assert np.array_equal(out, want.astype(np.float32))
    assert np.array_equal(out[0, 0], (d[0, 0, :3] / 16.0).astype(np.float32))


# ---- the device pass against the oracle -------------------------------------------------------------------------------------


@pytest.mark.gpu
def test_device_direct_pass_bitwise_on_analytic_scenes(binding, oracle):
    """iile_render_direct against oracle_iispt_direct, film monitor doubles bit for bit: a point light, an area light with
    MIS, UberMaterial (diffuse + a specular lobe whose Fresnel term is 0), the emitting mirror ball (four recursions)."""
    for name in ("furnace_point.pbrt", "furnace_area.pbrt", "furnace_uber.pbrt", "furnace_mirror_emitter.pbrt"):
        scene = binding.HostScene(path=os.path.join(SCENES, name))
        gpu = binding.GpuScene(scene)
        dev = gpu.render_direct(3)
        ref = oracle.iispt_direct(scene, 3)
        assert np.array_equal(dev.view(np.uint64), ref.view(np.uint64)), name


@pytest.mark.gpu
def test_device_direct_pass_bitwise_on_killeroo_and_rooms(binding, oracle, tmp_path):
    """killeroo-simple (sphere light, matte / plastic) and two rooms of tests/boxroom.py (triangle emitters with several
    lights: UniformSampleAllLights samples every one of them; uber / mirror blobs: the recursion), passes added in order; a
    second call continuing at pass 2 accumulates into the same monitor."""
    import boxroom
    scenes = [binding.HostScene(xres=160, yres=120, spp=1)]
    for i, kw in enumerate((dict(light="quad"), dict(light="multi", materials="mixed"), dict(light="spot", materials="mixed"))):
        path = tmp_path / f"room{i}.pbrt"
        path.write_text(boxroom.boxroom_pbrt(ico_levels=2, n_blobs=5, wall_n=6, xres=96, yres=64, spp=1, **kw))
        scenes.append(binding.HostScene(path=str(path)))
    for scene in scenes:
        gpu = binding.GpuScene(scene)
        dev = gpu.render_direct(3)
        ref = oracle.iispt_direct(scene, 3)
        assert np.array_equal(dev.view(np.uint64), ref.view(np.uint64))
        import torch
        h, w = scene.film_shape
        film = torch.zeros((h, w, 4), dtype=torch.float64, device="cuda")
        gpu.render_direct(2, film_device_ptr=film.data_ptr())
        gpu.render_direct(1, first_pass=2, film_device_ptr=film.data_ptr(), accumulate=True)
        torch.cuda.synchronize()
        assert np.array_equal(film.cpu().numpy().view(np.uint64), ref.view(np.uint64))


@pytest.mark.gpu
def test_direct_pass_rejects_what_it_does_not_build(binding, tmp_path):
    import boxroom
    path = tmp_path / "env.pbrt"
    path.write_text(boxroom.boxroom_pbrt(ico_levels=2, n_blobs=3, wall_n=4, xres=32, yres=32, spp=1, light="envmap", materials="mixed", textures=str(tmp_path)))
    gpu = binding.GpuScene(binding.HostScene(path=str(path)))
    with pytest.raises(RuntimeError, match="not built|differentials"):
        gpu.render_direct(1)
