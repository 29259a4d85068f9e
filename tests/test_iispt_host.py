"""The IISPT integrator's C++ host (csrc/host/gpu_iispt_integrator.h behind `iile_pbrt`; IISPTIntegrator::render_normal_2,
src/integrators/iispt.cpp:357-446) and the film-monitor entry points it shares with the Python frame.

CPU: the loader takes `Integrator "iispt"`, the weights file round trip, the command line's errors before any device is touched.
GPU: iile_iispt_film_add against numpy's doubles, and the image `iile_pbrt --integrator iispt` writes against the Python frame's
(pbrt-v3-iile_amd/iispt_frame.py: the same C ABI calls from the other host language) bit for bit."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(REPO, "pbrt-v3-iile_amd", "lib", "iile_pbrt")
KILLEROO = os.path.join(REPO, "scenes", "killeroo-simple.pbrt")

TENSORS = {  # IISPTNet's state_dict (ml/iispt_net.py:27-88): name -> shape
    "encoder0.0": (64, 7, 3, 3), "encoder0.2": (64, 64, 3, 3), "encoder1.1": (128, 64, 3, 3), "encoder1.4": (128, 128, 3, 3),
    "encoder2.1": (256, 128, 3, 3), "encoder2.4": (256, 256, 3, 3), "encoder3.1": (512, 256, 3, 3), "encoder3.4": (256, 512, 3, 3),
    "decoder0.0": (512, 256, 3, 3), "decoder0.3": (256, 128, 3, 3), "decoder1.0": (256, 128, 3, 3), "decoder1.3": (128, 64, 3, 3),
    "decoder2.0": (128, 64, 3, 3), "decoder2.2": (64, 64, 3, 3), "decoder2.4": (3, 64, 1, 1),
}


DECONVS = ("decoder0.0", "decoder0.3", "decoder1.0", "decoder1.3", "decoder2.0", "decoder2.2")


def _random_state(seed):
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in TENSORS.items():
        sd[name + ".weight"] = rng.normal(0, 0.05, shape).astype(np.float32)
        # ConvTranspose2d (decoder0.0 .. decoder2.2) keeps [in][out][k][k]
        sd[name + ".bias"] = rng.normal(0, 0.05, shape[1] if name in DECONVS else shape[0]).astype(np.float32)
    return sd


def test_loader_takes_the_iispt_integrator(binding, tmp_path):
    """MakeIntegrator's choice (src/core/api.cpp:1720-1750) is the file's Integrator line: the host reports it."""
    text = open(KILLEROO).read()
    assert 'Integrator "path"' in text
    scene = tmp_path / "iispt.pbrt"
    scene.write_text(text.replace('Integrator "path"', 'Integrator "iispt" "integer maxdepth" [7]').replace("geometry/", os.path.join(REPO, "scenes", "geometry") + "/"))
    a = binding.HostScene(path=str(scene), xres=64, yres=48, spp=1)
    b = binding.HostScene(xres=64, yres=48, spp=1)
    assert a.info["integrator"] == 1 and a.info["max_depth"] == 7 and b.info["integrator"] == 0
    bad = tmp_path / "bad.pbrt"
    bad.write_text(text.replace('Integrator "path"', 'Integrator "bdpt"'))
    with pytest.raises(RuntimeError, match='only Integrator "path" and "iispt"'):
        binding.HostScene(path=str(bad))


def test_weights_file_layout(binding, tmp_path):
    from_binding = {k: TENSORS[k] for k in binding.NET_CONVS}
    assert list(from_binding) == list(TENSORS)
    sd = _random_state(1)
    for i, k in enumerate(binding.NET_BNS):
        c = {"encoder1.3": 128, "encoder2.3": 256, "encoder3.3": 512, "decoder0.2": 512, "decoder1.2": 256}[k]
        for q in ("weight", "bias", "running_mean", "running_var"):
            sd[f"{k}.{q}"] = np.full(c, 1.0 + i, np.float32)
    path = tmp_path / "net.iilenet"
    binding.save_net_weights(sd, str(path), bn_eps=1e-5)
    raw = path.read_bytes()
    n_floats = sum(int(np.prod(s)) for s in TENSORS.values()) + sum(sd[k + ".bias"].size for k in TENSORS) + 4 * (128 + 256 + 512 + 512 + 256)
    assert raw[:8] == b"IILENET1" and len(raw) == 12 + 4 * n_floats
    assert np.frombuffer(raw[8:12], "<f4")[0] == np.float32(1e-5)
    assert np.array_equal(np.frombuffer(raw[12:12 + 4 * 64 * 7 * 9], "<f4"), sd["encoder0.0.weight"].reshape(-1))
    # a file of other tensors is refused while it is read — before any device is asked for
    lib = binding.gpu_lib()
    import ctypes
    for blob, what in ((raw[:-4], "short"), (raw + b"\0\0\0\0", "long"), (b"IILENET2" + raw[8:], "magic")):
        p = tmp_path / f"{what}.iilenet"
        p.write_bytes(blob)
        h = ctypes.c_void_p()
        assert lib.iile_iispt_net_load(os.fsencode(str(p)), ctypes.byref(h)) != 0 and not h
        assert b"not an IILENET1 file" in lib.iile_last_error()
    h = ctypes.c_void_p()
    assert lib.iile_iispt_net_load(os.fsencode(str(tmp_path / "missing")), ctypes.byref(h)) != 0 and b"cannot open" in lib.iile_last_error()


def test_cli_refuses_an_iispt_frame_it_cannot_render(tmp_path):
    """Errors of the IISPT branch that need no device: no weights, devices that are not there, --gpus with --gpurank, a probe side the
    network does not take."""
    def run(*args, env=None):
        e = dict(os.environ)
        e.pop("IILE_IISPT_NET", None)
        e.update(env or {})
        return subprocess.run([EXE, KILLEROO, "--xres", "32", "--yres", "24", "--spp", "1", "--outfile", str(tmp_path / "o.pfm"), *args],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=e)
    p = run("--integrator", "iispt")
    assert p.returncode == 1 and "needs the network's weights" in p.stdout
    p = run("--integrator", "iispt", "--gpus", "2", "--iisptNet=x")   # (one process, two devices: none is there)
    assert p.returncode == 1 and "given up on every device" in p.stdout
    p = run("--integrator", "iispt", "--gpus", "2", "--gpurank", "0/2", "--rendezvous", str(tmp_path / "rv"), "--iisptNet=x")
    assert p.returncode == 1
    p = run("--integrator", "iispt", "--iisptNet=x", "--iispt_hemi_size=16")
    assert p.returncode == 1 and "32 x 32 probes" in p.stdout
    p = run("--integrator", "bdpt")
    assert p.returncode == 1 and "--integrator wants path or iispt" in p.stdout
    assert not (tmp_path / "o.pfm").exists()


def test_cpp_schedule_is_the_python_schedule(tmp_path):
    """IisptSchedule (C++) and iispt_frame.schedule (Python) are IisptScheduleMonitor::next_task (iisptschedulemonitor.cpp:40-79):
    the same tasks, the float radius and its floor included, over many sweeps and with the reference's environment overrides."""
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    exe = tmp_path / "schedule_probe"
    subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(REPO, "tests", "cpp", "schedule_probe.cpp"), "-o", str(exe)], check=True, timeout=300)
    for bounds, n, start, ratio in (((0, 0, 1920, 1080), 3000, None, None), ((0, 0, 96, 80), 400, "4", None), ((-2, -2, 1282, 722), 900, "37.5", None),
                                    ((0, 0, 700, 700), 500, "100", "0.5"), ((0, 0, 33, 17), 300, "1.9", None)):
        env = {k: v for k, v in os.environ.items() if not k.startswith("IISPT_SCHEDULE_")}
        kw = {}
        if start is not None:
            env["IISPT_SCHEDULE_RADIUS_START"] = start
            kw["radius_start"] = float(np.float32(start))
        if ratio is not None:
            env["IISPT_SCHEDULE_RADIUS_RATIO"] = ratio
            kw["update_multiplier"] = float(np.float32(ratio))
        out = subprocess.run([str(exe), *map(str, bounds), str(n)], stdout=subprocess.PIPE, text=True, check=True, env=env, timeout=60).stdout
        got = [tuple(int(v) for v in line.split()) for line in out.splitlines()]
        want = list(frame_mod.schedule(bounds, n, **kw))
        assert len(got) == n and [g[:5] for g in got] == want, (bounds, start, ratio)
        assert [g[6] for g in got] == list(range(n))
        # `pass` counts the sweeps: it steps exactly where a task starts at the bounds' corner again
        assert all((g[5] != p[5]) == (g[:2] == bounds[:2]) for p, g in zip(got, got[1:]))
        assert got[-1][4] >= 1


def _read_pfm(path, w, h):
    raw = open(path, "rb").read()
    head = f"PF\n{w} {h}\n-1.0\n".encode()
    assert raw.startswith(head)
    return np.frombuffer(raw[len(head):], "<f4").reshape(h, w, 3)[::-1]


@pytest.mark.gpu
def test_film_add_matches_numpy(binding):
    """IisptFilmMonitor::add_n_samples over the tasks of a sweep in one launch: float samples added to double sums; a second sweep
    (other rectangles over the same pixels) on top."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    scene = binding.HostScene(xres=96, yres=80, spp=1)
    gpu = binding.GpuScene(scene)
    film = torch.zeros((80, 96, 4), dtype=torch.float64, device="cuda")
    ref = np.zeros((80, 96, 4), np.float64)
    rng = np.random.default_rng(11)
    tasks = list(frame_mod.schedule((0, 0, 96, 80), 18, radius_start=4.0))
    for tilesize in (4, 3):
        sweep = [binding.IisptTask(x0, y0, x1, y1, ts, 0, 0) for (x0, y0, x1, y1, ts) in tasks if ts == tilesize]
        assert sweep
        n_pix = sum((t.x1 - t.x0) * (t.y1 - t.y0) for t in sweep)
        out = (rng.uniform(0, 3, (n_pix, 4)) * rng.choice([1e-6, 1.0, 1e5], (n_pix, 1))).astype(np.float32)
        at = 0
        for t in sweep:
            n = (t.x1 - t.x0) * (t.y1 - t.y0)
            ref[t.y0:t.y1, t.x0:t.x1] += out[at:at + n].reshape(t.y1 - t.y0, t.x1 - t.x0, 4).astype(np.float64)
            at += n
        out_t = torch.from_numpy(out).cuda()
        gpu.iispt_film_add(sweep, out_t.data_ptr(), film.data_ptr())
        torch.cuda.synchronize()
    assert (ref[..., 3] > 0).all()
    assert np.array_equal(film.cpu().numpy().view(np.uint64), ref.view(np.uint64))
    with pytest.raises(RuntimeError, match="outside the film"):
        gpu.iispt_film_add([binding.IisptTask(90, 0, 100, 10, 4, 0, 0)], out_t.data_ptr(), film.data_ptr())


@pytest.mark.gpu
def test_cpp_iispt_integrator_writes_the_python_frames_image(binding, tmp_path):
    """`iile_pbrt --integrator iispt` (GpuIisptIntegrator: schedule, groups, probe pass, network from an IILENET1 file, gather, film
    monitors, direct pass, merge — all through the C ABI) against IisptFrame doing the same from Python with the network built from
    the state_dict: the three images the reference writes (indirect, direct, merged), bit for bit."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    sys.path.insert(0, REPO)
    nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
    import iispt_torch_reference as ref_mod
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    w, h, n_tasks, n_direct = 96, 80, 21, 3
    torch.manual_seed(3)
    module = ref_mod.IISPTNet().eval()
    net_file = tmp_path / "net.iilenet"
    binding.save_net_weights(module.state_dict(), str(net_file), bn_eps=module.encoder1[3].eps)
    out, ind, direct = tmp_path / "frame.pfm", tmp_path / "indirect.pfm", tmp_path / "direct.pfm"
    env = dict(os.environ, IISPT_SCHEDULE_RADIUS_START="4")
    p = subprocess.run([EXE, KILLEROO, "--xres", str(w), "--yres", str(h), "--spp", "1", "--integrator", "iispt", f"--iisptNet={net_file}",
                        f"--iileIndirect={n_tasks}", f"--iileDirect={n_direct}", "--outfile", str(out), f"--iisptIndirectOut={ind}",
                        f"--iisptDirectOut={direct}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout
    scene = binding.HostScene(xres=w, yres=h, spp=1)
    gpu = binding.GpuScene(scene)
    pipe = nn_mod.IisptPipeline(gpu, net=module)
    frame = frame_mod.IisptFrame(binding, gpu, pipe)
    frame.run_batched(n_tasks, radius_start=4.0)
    frame.run_direct(n_direct)
    torch.cuda.synchronize()
    assert f"IISPT: {n_tasks} tasks, {frame.stats['hemi_points']} hemi points, {frame.stats['probes']} probes, {frame.stats['pixels']} pixels" in p.stdout
    # two sweeps and the start of a third, whose radius floors to the second's 3 pixels: groups are cut at the sweeps, not where the
    # tile size changes (inside a sweep no two tasks share a pixel; the third sweep's first tasks cover pixels of the second's)
    assert [t[4] for t in frame_mod.schedule((0, 0, w, h), n_tasks, 4.0)] == [4] * 6 + [3] * 15
    for path, want, name in ((out, frame.image(), "merged"), (ind, frame.indirect_image(), "indirect"), (direct, frame.direct_image(), "direct")):
        got = _read_pfm(path, w, h)
        want = want.cpu().numpy()
        assert float(want.max()) > 0, name
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    # the network object made from the file is the one made from the state_dict
    a, b = binding.GpuNet(module.state_dict(), bn_eps=module.encoder1[3].eps), binding.GpuNet(path=str(net_file))
    x = torch.rand((5, 7, 32, 32), device="cuda")
    ya, yb = torch.empty((5, 3, 32, 32), device="cuda"), torch.empty((5, 3, 32, 32), device="cuda")
    a.forward(x.data_ptr(), ya.data_ptr(), 5)
    b.forward(x.data_ptr(), yb.data_ptr(), 5)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)


@pytest.mark.gpu
def test_frame_shards_add_up_to_the_frame(binding):
    """The IISPT frame over several GPUs (iispt_frame.py: rank / nranks): tasks dealt by their number, direct passes in contiguous
    blocks, counters and seeds advancing over ALL tasks — so every rank renders exactly the tasks and passes the single process
    would, and the ranks' film monitors (sums of doubles per pixel, IisptFilmMonitor::add_n_samples) add up to the frame's. The
    ranks are played one after the other on this GPU; their monitors added as the all-reduce would add them: indirect monitor,
    direct monitor and merged image equal to the single-rank frame's bit for bit, for 2, 3 and 5 ranks, over two sweeps and the start
    of a third (a pixel then belongs to tasks of different ranks)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    sys.path.insert(0, REPO)
    nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
    import iispt_torch_reference as ref_mod
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    w, h, n_tasks, n_direct = 96, 80, 21, 5
    torch.manual_seed(3)
    scene = binding.HostScene(xres=w, yres=h, spp=1)
    gpu = binding.GpuScene(scene)
    pipe = nn_mod.IisptPipeline(gpu, net=ref_mod.IISPTNet().eval())
    whole = frame_mod.IisptFrame(binding, gpu, pipe)
    whole.run_batched(n_tasks, radius_start=4.0)
    whole.run_direct(n_direct)
    want = whole.image().cpu().numpy()
    assert float(want.max()) > 0 and whole.stats["tasks"] == n_tasks
    for nranks in (2, 3, 5):
        parts = []
        for rank in range(nranks):
            f = frame_mod.IisptFrame(binding, gpu, pipe)
            f.run_batched(n_tasks, radius_start=4.0, rank=rank, nranks=nranks)
            f.run_direct(n_direct, rank=rank, nranks=nranks)
            parts.append(f)
        assert all(0 < f.stats["tasks"] < n_tasks for f in parts) and sum(f.stats["tasks"] for f in parts) == n_tasks
        # a rank's indirect monitor covers its own tasks only
        assert (parts[0].film[..., 3] > 0).sum() < (whole.film[..., 3] > 0).sum()
        total = parts[0].reduce_monitors(others=parts[1:])
        assert total.stats == whole.stats
        assert torch.equal(total.film, whole.film), nranks
        assert torch.equal(total.film_direct, whole.film_direct), nranks
        assert np.array_equal(total.image().cpu().numpy().view(np.uint32), want.view(np.uint32)), nranks


@pytest.mark.gpu
def test_cpp_iispt_frame_in_shards(binding, tmp_path):
    """The C++ IISPT host over several GPUs (`iile_pbrt --integrator iispt --gpurank R/N`: gpu_iispt_integrator.h rank / nranks /
    comm). (1) Through the communicator branch with ONE rank — all_ok, the two `iile_dist_monitor_reduce` (doubles over RCCL), the summed
    statistics — the three images are the plain CLI's byte for byte. (2) The share a rank renders is the share IisptFrame renders for
    that rank (whose shares add up to the frame: test_frame_shards_add_up_to_the_frame): the images of rank R of 3 alone
    ($IILE_DEBUG_IISPT_SHARD, no communicator) against the Python share's, bit for bit."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    sys.path.insert(0, REPO)
    nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
    import iispt_torch_reference as ref_mod
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    w, h, n_tasks, n_direct = 96, 80, 21, 5
    torch.manual_seed(3)
    module = ref_mod.IISPTNet().eval()
    net_file = tmp_path / "net.iilenet"
    binding.save_net_weights(module.state_dict(), str(net_file), bn_eps=module.encoder1[3].eps)

    def cli(tag, *extra, env=None):
        outs = [tmp_path / f"{tag}_{k}.pfm" for k in ("frame", "indirect", "direct")]
        e = dict(os.environ, IISPT_SCHEDULE_RADIUS_START="4")
        e.update(env or {})
        p = subprocess.run([EXE, KILLEROO, "--xres", str(w), "--yres", str(h), "--spp", "1", "--integrator", "iispt", f"--iisptNet={net_file}",
                            f"--iileIndirect={n_tasks}", f"--iileDirect={n_direct}", "--outfile", str(outs[0]), f"--iisptIndirectOut={outs[1]}",
                            f"--iisptDirectOut={outs[2]}", *extra], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=e)
        assert p.returncode == 0, p.stdout
        return outs, p.stdout

    plain, out0 = cli("plain")
    ranked, out1 = cli("ranked", "--gpurank", "0/1", "--rendezvous", str(tmp_path / "rv"), "--job", "5")
    for a, b_ in zip(plain, ranked):
        assert a.read_bytes() == b_.read_bytes()
    # one process, the devices of the node: `--gpus 1` runs GpuIisptIntegrator::RenderAllDevices — a host thread per device, an in-process
    # rendezvous, the same shares and monitor reduction — with one device: the same images; an injected set-up fault ends the run with
    # exit code 1 and no image instead of a hang
    threaded, _ = cli("threaded", "--gpus", "1")
    for a, b_ in zip(plain, threaded):
        assert a.read_bytes() == b_.read_bytes()
    bad = tmp_path / "bad.pfm"
    e = dict(os.environ, IISPT_SCHEDULE_RADIUS_START="4", IILE_DEBUG_GANG_FAULT="create:0", IILE_DIST_TIMEOUT_S="5")
    p = subprocess.run([EXE, KILLEROO, "--xres", str(w), "--yres", str(h), "--spp", "1", "--integrator", "iispt", f"--iisptNet={net_file}", f"--iileIndirect={n_tasks}",
                        "--gpus", "1", "--outfile", str(bad)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120, env=e)
    assert p.returncode == 1 and "given up on every device" in p.stdout and not bad.exists(), p.stdout
    assert [l for l in out0.splitlines() if l.startswith("IISPT:")][0].split("->")[0] == [l for l in out1.splitlines() if l.startswith("IISPT:")][0].split("->")[0]
    scene = binding.HostScene(xres=w, yres=h, spp=1)
    gpu = binding.GpuScene(scene)
    pipe = nn_mod.IisptPipeline(gpu, net=module)
    for rank in range(3):
        outs, _ = cli(f"shard{rank}", env={"IILE_DEBUG_IISPT_SHARD": f"{rank}/3"})
        f = frame_mod.IisptFrame(binding, gpu, pipe)
        f.run_batched(n_tasks, radius_start=4.0, rank=rank, nranks=3)
        f.run_direct(n_direct, rank=rank, nranks=3)
        torch.cuda.synchronize()
        for path, want, name in ((outs[0], f.image(), "merged"), (outs[1], f.indirect_image(), "indirect"), (outs[2], f.direct_image(), "direct")):
            got = _read_pfm(path, w, h)
            assert np.array_equal(got.view(np.uint32), want.cpu().numpy().view(np.uint32)), (rank, name)
