"""The network stage of the IISPT probe pipeline (pbrt-v3-iile_amd/iispt_nn.py): the two transforms against a
scalar restatement of the reference's ImageFilm arithmetic, the network's checkpoint layout, and (GPU) the
device-resident pipeline render -> normalise -> network -> rescale."""
import importlib
import math

import numpy as np
import pytest
import torch

nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")   # the product: HIP only
import iispt_torch_reference as ref_mod                          # the PyTorch statement it is held against (tests/)


def _downstream_scalar(inten, nrm, dist):
    """normalizeMapsDownstream (iisptrenderrunner.cpp:1041-1092) over ImageFilm's operations
    (imagefilm.cpp:203-254, 298-379), one probe, plain Python floats rounded to float32 where the reference holds
    a float."""
    f32 = np.float32
    h = inten.shape[0]
    chan = [f32(sum(float(v) for v in inten[..., c].ravel()) / (h * h)) for c in range(3)]
    s, cnt = 0.0, 0
    for px in inten.reshape(-1, 3):          # computeMean: sum += (r + g + b) in float, accumulated in double
        s += float(f32(f32(px[0] + px[1]) + px[2]))
        cnt += 3
    mean = f32(s / cnt)
    ratio = f32(0.0) if mean == 0 else f32(1.0 / (10.0 * float(mean)))
    out = np.zeros((h, h, 7), np.float32)
    for y in range(h):
        for x in range(h):
            for c in range(3):
                v = f32(inten[y, x, c] * ratio)
                v = f32(math.log(1.0 + (float(v) if v > 0 else 0.0)))
                out[y, x, c] = f32(v + f32(-0.1))
                n = f32(f32(nrm[y, x, c] - f32(0)) / f32(1))
                out[y, x, 3 + c] = f32(-1) if n < -1 else (f32(1) if n > 1 else n)
    zmean = f32(sum(float(v) for v in dist.ravel()) / (h * h))
    div = f32(10.0 * (float(zmean) + 1.0))
    if div == 0:
        div = f32(1)
    r = f32(1.0 / float(div))
    for y in range(h):
        for x in range(h):
            v = f32(f32(dist[y, x] + f32(1.0)) * r)
            v = f32(math.log(1.0 + (float(v) if v > 0 else 0.0)))
            out[y, x, 6] = f32(v + f32(-0.1))
    return out[::-1].transpose(2, 0, 1), np.array(chan, np.float32)   # ImageFilm rows, (channel, height, width)


def test_transforms_follow_the_reference_arithmetic():
    rng = np.random.default_rng(0)
    n, h = 3, 32
    inten = (rng.random((n, h, h, 3)) ** 3 * 5).astype(np.float32)
    inten[1] = 0                                                      # a black probe: ratio 0
    nrm = rng.uniform(-1.2, 1.2, (n, h, h, 3)).astype(np.float32)     # a little outside [-1, 1]: clamped
    dist = rng.uniform(0, 40, (n, h, h)).astype(np.float32)
    dist[2, :10] = -1                                                 # escaped rays
    x, means = ref_mod.normalize_downstream(torch.from_numpy(inten), torch.from_numpy(nrm), torch.from_numpy(dist))
    assert x.shape == (n, 7, h, h) and means.shape == (n, 3)
    for i in range(n):
        want, chan = _downstream_scalar(inten[i], nrm[i], dist[i])
        assert np.allclose(means[i].numpy(), chan, rtol=1e-6)
        # computeMean adds r + g + b in float first: the batched mean differs from it by rounding only
        assert np.allclose(x[i].numpy(), want, rtol=0, atol=3e-6)
    # upstream: exp(max(v, 0)) - 1, then every channel rescaled to the rendered probe's mean; rows flipped back
    out = torch.from_numpy(rng.uniform(-0.2, 2.0, (n, 3, h, h)).astype(np.float32))
    y = ref_mod.transform_upstream(out, means).numpy()
    assert y.shape == (n, h, h, 3)
    for i in range(n):
        e = np.exp(np.maximum(out[i].numpy().astype(np.float64), 0)) - 1
        for c in range(3):
            actual = np.float32(e[c].astype(np.float32).astype(np.float64).mean())
            mul = means[i, c].item() / actual if actual > 1e-10 else 0.0
            assert np.allclose(y[i, ::-1, :, c], e[c].astype(np.float32) * np.float32(mul), rtol=2e-6, atol=1e-7)
            if means[i, c] > 0:
                assert abs(float(y[i, ..., c].astype(np.float64).mean()) - float(means[i, c])) < 1e-5 * float(means[i, c]) + 1e-7
    assert (y[1] == 0).all()                                           # the black probe stays black


def test_network_has_the_reference_checkpoint_layout():
    """ml/iispt_net.py:8-109: parameter names and shapes as `torch.save(net.state_dict())` of the reference writes
    them (K = 64), 7 -> 3 channels at 32 x 32, non-negative output (final ReLU), batch-independent in eval mode."""
    net = ref_mod.IISPTNet().eval()
    sd = net.state_dict()
    K = 64
    convs = {"encoder0.0": (K, 7, 3), "encoder0.2": (K, K, 3), "encoder1.1": (2 * K, K, 3), "encoder1.4": (2 * K, 2 * K, 3),
             "encoder2.1": (4 * K, 2 * K, 3), "encoder2.4": (4 * K, 4 * K, 3), "encoder3.1": (8 * K, 4 * K, 3),
             "encoder3.4": (4 * K, 8 * K, 3), "decoder2.4": (3, K, 1)}
    deconvs = {"decoder0.0": (8 * K, 4 * K), "decoder0.3": (4 * K, 2 * K), "decoder1.0": (4 * K, 2 * K), "decoder1.3": (2 * K, K),
               "decoder2.0": (2 * K, K), "decoder2.2": (K, K)}
    bns = {"encoder1.3": 2 * K, "encoder2.3": 4 * K, "encoder3.3": 8 * K, "decoder0.2": 4 * K, "decoder1.2": 2 * K}
    want = {}
    for k, (co, ci, ks) in convs.items():
        want[k + ".weight"], want[k + ".bias"] = (co, ci, ks, ks), (co,)
    for k, (ci, co) in deconvs.items():   # ConvTranspose2d keeps (in, out, k, k)
        want[k + ".weight"], want[k + ".bias"] = (ci, co, 3, 3), (co,)
    for k, c in bns.items():
        for p in ("weight", "bias", "running_mean", "running_var"):
            want[f"{k}.{p}"] = (c,)
        want[k + ".num_batches_tracked"] = ()
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    torch.manual_seed(1)
    x = torch.randn(5, 7, 32, 32)
    with torch.no_grad():
        y = net(x)
        assert y.shape == (5, 3, 32, 32) and (y >= 0).all()
        assert torch.allclose(net(x[2:3]), y[2:3], atol=1e-5)


def _fixture():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "iispt_net_fixture.npz"))


def _recipe_net():
    import iispt_net_recipe as recipe
    net = ref_mod.IISPTNet()
    state = recipe.fill_state_dict(net)
    return net.eval(), state


def test_network_reproduces_the_reference_modules_forward():
    """tests/golden/iispt_net_fixture.npz was made by the REFERENCE's `IISPTNet` (ml/iispt_net.py:8-109, imported in the build
    container by tests/golden/make_iispt_net_fixture.py): its state_dict entry names and shapes, the SHA-256 of every recipe
    tensor, a seeded (4, 7, 32, 32) input and the eval-mode output. This module must have the same entries in the same
    order, regenerate the same weights bit for bit and reproduce the output to 1e-5."""
    fx = _fixture()
    net, state = _recipe_net()
    assert [s[0] for s in state] == [str(x) for x in fx["state_names"]]
    assert [",".join(map(str, s[1])) for s in state] == [str(x) for x in fx["state_shapes"]]
    assert [s[2] for s in state] == [str(x) for x in fx["state_sha256"]]
    # BatchNorm statistics are off their defaults, so eval mode really uses them
    assert float(net.encoder1[3].running_var.min()) >= 0.5 and float(net.encoder1[3].running_mean.abs().max()) > 0.05
    with torch.no_grad():
        y = net(torch.from_numpy(fx["input"])).numpy()
    want = fx["output"]
    assert want.shape == (4, 3, 32, 32) and (want > 0).mean() > 0.3
    assert np.abs(y - want).max() <= 1e-5 * max(1.0, float(np.abs(want).max())), float(np.abs(y - want).max())


def test_wire_order_is_the_references():
    """ml/main_stdio_net.py:47-86 run on seeded bytes (fixture): `read_input` -> the (7, h, w) array the network sees,
    `output_to_stdout` -> the floats on the pipe. The module's two layout helpers must give exactly those."""
    fx = _fixture()
    h = 32
    w = fx["wire_in"]
    inten = torch.from_numpy(w[: h * h * 3].reshape(1, h, h, 3))
    nrm = torch.from_numpy(w[h * h * 3: h * h * 6].reshape(1, h, h, 3))
    dist = torch.from_numpy(w[h * h * 6:].reshape(1, h, h))
    got = ref_mod.wire_to_network_input(inten, nrm, dist)[0].numpy()
    assert got.shape == (7, h, h) and np.array_equal(got, fx["wire_net_input"])
    back = ref_mod.network_output_to_wire(torch.from_numpy(fx["wire_net_output"]).unsqueeze(0))[0].contiguous().numpy()
    assert np.array_equal(back.ravel(), fx["wire_out"])


def per_element(got, want):
    """north_star's tolerance, per element (VERDICT r05 "next" 2): the share of elements with
    |err| <= 1e-4 |want| + 1e-6 max|want|, the mean relative error over the elements that are not zero, max|err| / max|want|."""
    got, want = np.asarray(got, np.float64).ravel(), np.asarray(want, np.float64).ravel()
    mx = float(np.abs(want).max())
    err = np.abs(got - want)
    nz = np.abs(want) > 1e-6 * mx
    return float((err <= 1e-4 * np.abs(want) + 1e-6 * mx).mean()), float((err[nz] / np.abs(want[nz])).mean()), float(err.max() / mx)


def assert_per_element(got, want, what):
    inside, mean_rel, max_over_max = per_element(got, want)
    print(f"{what}: {inside * 100:.4f} % of elements within 1e-4 relative (+ 1e-6 of the maximum), mean relative error {mean_rel:.2e}, max error / max {max_over_max:.2e}")
    assert inside >= 0.999 and mean_rel <= 1e-5, (what, inside, mean_rel, max_over_max)
    return inside, mean_rel, max_over_max


# the module whose OUTPUT is what convolution layer l of csrc/device/iispt_net.hip stores (behind LeakyReLU / BatchNorm)
_LAYER_TAPS = [("encoder0", 1), ("encoder0", 3), ("encoder1", 3), ("encoder1", 5), ("encoder2", 3), ("encoder2", 5), ("encoder3", 3),
               ("encoder3", 5), ("decoder0", 2), ("decoder0", 4), ("decoder1", 2), ("decoder1", 4), ("decoder2", 1), ("decoder2", 3)]


@pytest.mark.gpu
def test_network_on_the_gpu_against_the_reference_fixture(binding):
    """The product's network — the hand-written kernels behind iile_iispt_net_* (split-fp16 matrix instructions: 22 significant
    bits per operand, fp32 accumulation) — against the REFERENCE module's output (fixture): PER ELEMENT inside north_star's 1e-4
    relative band (>= 99.9 % of elements, mean relative error <= 1e-5) and within 1e-5 of the largest value everywhere. Beside it,
    for the record, eager PyTorch on the same device in fp32 and in plain bf16 (neither is the product path)."""
    torch.cuda.init()
    fx = _fixture()
    net, _ = _recipe_net()
    x = torch.from_numpy(fx["input"]).cuda()
    want = fx["output"]
    scale = float(np.abs(want).max())
    g = binding.GpuNet(net.state_dict())
    y = torch.empty((4, 3, 32, 32), dtype=torch.float32, device="cuda")
    g.forward(x.data_ptr(), y.data_ptr(), 4)
    torch.cuda.synchronize()
    err_hip = float(np.abs(y.cpu().numpy() - want).max()) / scale
    with torch.no_grad():
        y32 = net.cuda()(x.contiguous(memory_format=torch.channels_last)).float().cpu().numpy()
        y16 = net.to(torch.bfloat16)(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)).float().cpu().numpy()
    err32 = float(np.abs(y32 - want).max()) / scale
    err16 = float(np.abs(y16 - want).max()) / scale
    print(f"IISPTNet vs the reference fixture, max error over the largest output: HIP kernels {err_hip:.2e}; eager PyTorch fp32 {err32:.2e}, bf16 {err16:.2e}")
    assert err_hip < 1e-5, err_hip      # (round 5's split-bf16 kernels: 2.4e-5, and 96 % of the elements inside the band)
    assert_per_element(y.cpu().numpy(), want, "HIP network vs the reference module's fixture (4 probes)")
    assert err32 < 1e-4, err32
    assert err16 < 0.1, err16    # measured, not a parity claim: plain bf16 keeps 8 significant bits through 15 convolutions


@pytest.mark.gpu
def test_hip_network_layer_by_layer_against_the_module(binding):
    """Every convolution layer of the HIP network (with the pooling / upsampling + concatenation folded into its loads and
    the activation / BatchNorm folded into its stores) against the same point of the PyTorch module on the CPU, on a batch
    that fills no tile exactly (37 probes: tiles hold 1, 4 or 16 images); then the batch cut into pieces of 10 must give the
    same bits, and so must a batch of one."""
    torch.cuda.init()
    import iispt_net_recipe as recipe
    net, _ = _recipe_net()
    n = 37
    xin = torch.from_numpy(recipe.fixture_input(n))
    acts, hooks = {}, []
    for l, (blk, idx) in enumerate(_LAYER_TAPS):
        hooks.append(getattr(net, blk)[idx].register_forward_hook(lambda m, i, o, l=l: acts.__setitem__(l, o.detach())))
    with torch.no_grad():
        ref = net(xin)
    for h in hooks:
        h.remove()
    g = binding.GpuNet(net.state_dict())
    xd = xin.cuda()
    yd = torch.empty((n, 3, 32, 32), dtype=torch.float32, device="cuda")
    for l in range(14):
        a = acts[l]
        lo = torch.empty((n, a.shape[2], a.shape[3], a.shape[1]), dtype=torch.float32, device="cuda")
        g.forward(xd.data_ptr(), yd.data_ptr(), n, layer_out_ptr=lo.data_ptr(), layer=l)
        torch.cuda.synchronize()
        err = float((lo.cpu().permute(0, 3, 1, 2) - a).abs().max() / a.abs().max())
        assert err < 5e-6, (l, err)
    assert float((yd.cpu() - ref).abs().max() / ref.abs().max()) < 1e-5
    y2 = torch.empty_like(yd)
    g.forward(xd.data_ptr(), y2.data_ptr(), n, max_batch=10)
    y1 = torch.empty((1, 3, 32, 32), dtype=torch.float32, device="cuda")
    g.forward(xd[5:6].contiguous().data_ptr(), y1.data_ptr(), 1)
    torch.cuda.synchronize()
    assert torch.equal(yd, y2) and torch.equal(yd[5:6], y1)
    g.forward(xd.data_ptr(), y2.data_ptr(), 0)   # nothing to do is not an error
    with pytest.raises(RuntimeError):
        g.forward(0, y2.data_ptr(), 3)


@pytest.mark.gpu
def test_hip_network_follows_a_checkpoint_not_the_recipe(binding):
    """Weights of another kind (PyTorch's default initialisation, BatchNorm statistics moved off their defaults): the HIP
    network built from that state_dict agrees with the module — the packer reads the checkpoint, not a fixed recipe."""
    torch.cuda.init()
    torch.manual_seed(11)
    net = ref_mod.IISPTNet().eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.3, 0.3)
                m.running_var.uniform_(0.4, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
        net.decoder2[4].bias.fill_(0.5)
    x = torch.randn(9, 7, 32, 32)
    with torch.no_grad():
        ref = net(x)
    assert float(ref.max()) > 0
    g = binding.GpuNet(net.state_dict())
    y = torch.empty((9, 3, 32, 32), dtype=torch.float32, device="cuda")
    g.forward(x.cuda().data_ptr(), y.data_ptr(), 9)
    torch.cuda.synchronize()
    assert float((y.cpu() - ref).abs().max() / ref.abs().max()) < 1e-5
    assert_per_element(y.cpu().numpy(), ref.numpy(), "HIP network vs the module with a foreign checkpoint (9 probes)")


@pytest.mark.gpu
def test_network_and_predicted_hemispheres_per_element_on_256_probes(binding):
    """north_star's tolerance where it is hardest to meet (VERDICT r05 "next" 2): 256 probes — rendered ones, of killeroo-simple —
    through iile_iispt_net_forward and through iile_iispt_net_predict, against the fp32 module on the CPU and the tensor-expression
    transforms, PER ELEMENT: |err| <= 1e-4 |want| + 1e-6 max on >= 99.9 % of the elements, mean relative error <= 1e-5 — for the
    network's output and for the predicted hemispheres AFTER transformMapsUpstream (exp(v) - 1 and the per-channel rescale:
    iisptrenderrunner.cpp:1095-1133). A three-product split of bf16 halves (round 5) reaches 96 % / 7e-5 here; fp16 halves with
    the weights' exponents moved into the normal range reach the fp32 module's own distance from exact arithmetic."""
    torch.cuda.init()
    scene = binding.HostScene(xres=64, yres=64, spp=1)
    gpu = binding.GpuScene(scene)
    rng = np.random.default_rng(17)
    n = 256
    pos = rng.uniform((-150, -100, -130), (250, 150, 0), (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    net, _ = _recipe_net()
    pipe = nn_mod.IisptPipeline(gpu, net=net, binding=binding)
    pred, inten, nrm, dist = pipe(pos, d, batch=100)
    with torch.no_grad():
        x, means = ref_mod.normalize_downstream(inten.cpu(), nrm.cpu(), dist.cpu())
        y_ref = net(x)
        want = ref_mod.transform_upstream(y_ref, means)
        y_hip = pipe.infer(x.cuda()).cpu()
    assert bool(torch.isfinite(pred).all()) and float(want.abs().max()) > 0
    assert_per_element(y_hip.numpy(), y_ref.numpy(), "network output, 256 rendered probes, HIP vs the fp32 module on the CPU")
    assert_per_element(pred.cpu().numpy(), want.numpy(), "predicted hemispheres after transformMapsUpstream, 256 probes")


@pytest.mark.gpu
def test_a_non_finite_probe_value_does_not_poison_the_prediction(binding):
    """an inf / NaN in a probe image (the probe pass guards its radiance, so none is expected) used to turn the whole
    256-pixel tile's sums into NaN through inf - inf in the operand split. k_net_normalize now counts such a value as 0 — the
    film's own rule for such samples — and the staged operands are clamped into fp16's range: the prediction is finite, and equal
    to the prediction of the same probe with zeros in those places; a huge activation saturates instead of becoming inf."""
    torch.cuda.init()
    rng = np.random.default_rng(3)
    n, h = 5, 32
    inten = (rng.random((n, h, h, 3)) * 2).astype(np.float32)
    nrm = rng.uniform(-1, 1, (n, h, h, 3)).astype(np.float32)
    dist = rng.uniform(0, 40, (n, h, h)).astype(np.float32)
    bad_i, bad_n, bad_d = inten.copy(), nrm.copy(), dist.copy()
    bad_i[1, 3, 4, 0] = np.inf
    bad_i[1, 9, 9, 2] = np.nan
    bad_d[2, 0, 0] = np.inf
    bad_n[3, 5, 5, 1] = np.nan
    zero_i, zero_n, zero_d = inten.copy(), nrm.copy(), dist.copy()
    zero_i[1, 3, 4, 0] = 0
    zero_i[1, 9, 9, 2] = 0
    zero_d[2, 0, 0] = 0
    zero_n[3, 5, 5, 1] = 0
    net, _ = _recipe_net()
    g = binding.GpuNet(net.state_dict())
    outs = []
    for a, b_, c in ((bad_i, bad_n, bad_d), (zero_i, zero_n, zero_d)):
        da, db, dc = (torch.from_numpy(v).cuda() for v in (a, b_, c))
        out = torch.empty((n, h, h, 3), dtype=torch.float32, device="cuda")
        g.predict(da.data_ptr(), db.data_ptr(), dc.data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1])
    # the bare network with an input far outside fp16's range: finite output (saturated operands), the other probes untouched
    x = torch.from_numpy(rng.standard_normal((3, 7, h, h)).astype(np.float32))
    x_big = x.clone()
    x_big[1, 0, 5, 5] = 3e7
    ya, yb = (torch.empty((3, 3, h, h), dtype=torch.float32, device="cuda") for _ in range(2))
    g.forward(x.cuda().data_ptr(), ya.data_ptr(), 3)
    g.forward(x_big.cuda().data_ptr(), yb.data_ptr(), 3)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(yb).all()) and torch.equal(ya[0], yb[0]) and torch.equal(ya[2], yb[2])


@pytest.mark.gpu
def test_network_batch_shrinks_when_the_workspace_does_not_fit(binding, monkeypatch):
    """the activation workspace is 1.19 MiB per probe of a batch. When the device cannot hold the batch asked for
    (here: a 40 MiB cap through IILE_NET_WORKSPACE_MB) the batch is halved until it fits — same bits, more launches — and below 64
    probes the call fails with a message instead of crashing."""
    torch.cuda.init()
    import iispt_net_recipe as recipe
    net, _ = _recipe_net()
    n = 150
    x = torch.from_numpy(recipe.fixture_input(n)).cuda()
    g = binding.GpuNet(net.state_dict())
    y = torch.empty((n, 3, 32, 32), dtype=torch.float32, device="cuda")
    g.forward(x.data_ptr(), y.data_ptr(), n)
    torch.cuda.synchronize()
    monkeypatch.setenv("IILE_NET_WORKSPACE_MB", "40")     # room for 33 probes: 150 -> 75 -> 38 -> 19
    g2 = binding.GpuNet(net.state_dict())
    y2 = torch.empty_like(y)
    g2.forward(x.data_ptr(), y2.data_ptr(), n)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    monkeypatch.setenv("IILE_NET_WORKSPACE_MB", "1")
    g3 = binding.GpuNet(net.state_dict())
    with pytest.raises(RuntimeError, match="no room for the activations"):
        g3.forward(x.data_ptr(), y2.data_ptr(), n)


@pytest.mark.gpu
def test_predict_runs_the_two_transforms_as_the_tensor_expressions_do(binding):
    """iile_iispt_net_predict = normalizeMapsDownstream -> network -> transformMapsUpstream in three kernels around the
    convolutions. Against the same three steps as tensor expressions (normalize_downstream / transform_upstream above, held to a
    scalar restatement of the reference's ImageFilm arithmetic by test_transforms_follow_the_reference_arithmetic) with the
    SAME network kernels in between: equal up to the order of the double sums and one ulp of log / exp (1e-5 of the largest
    value); a black probe stays black, escaped rays (distance -1) and out-of-range normals are clamped, and film_rows returns
    the rows in the network's own order."""
    torch.cuda.init()
    rng = np.random.default_rng(7)
    n, h = 21, 32
    inten = (rng.random((n, h, h, 3)) ** 3 * 5).astype(np.float32)
    inten[1] = 0
    nrm = rng.uniform(-1.2, 1.2, (n, h, h, 3)).astype(np.float32)
    dist = rng.uniform(0, 40, (n, h, h)).astype(np.float32)
    dist[2, :10] = -1
    net, _ = _recipe_net()
    g = binding.GpuNet(net.state_dict())
    di, dn, dd = (torch.from_numpy(a).cuda() for a in (inten, nrm, dist))
    with torch.no_grad():
        x, means = ref_mod.normalize_downstream(di, dn, dd)
        x = x.contiguous()
        y = torch.empty((n, 3, h, h), dtype=torch.float32, device="cuda")
        g.forward(x.data_ptr(), y.data_ptr(), n)
        want = ref_mod.transform_upstream(y, means)
    got = torch.empty((n, h, h, 3), dtype=torch.float32, device="cuda")
    g.predict(di.data_ptr(), dn.data_ptr(), dd.data_ptr(), got.data_ptr(), n)
    rows = torch.empty_like(got)
    g.predict(di.data_ptr(), dn.data_ptr(), dd.data_ptr(), rows.data_ptr(), n, film_rows=True, max_batch=8)
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert scale > 0 and float((got - want).abs().max()) <= 1e-5 * scale, float((got - want).abs().max()) / scale
    assert bool((got[1] == 0).all())
    assert torch.equal(rows, torch.flip(got, dims=(1,)))
    # every channel of a predicted hemisphere has the mean of its rendered probe (where the network leaves it above zero)
    gm = got.double().reshape(n, -1, 3).mean(1).cpu().numpy()
    lit = (means.cpu().numpy() > 1e-6) & (gm > 0)
    assert lit.sum() > n and np.allclose(gm[lit], means.cpu().numpy()[lit], rtol=1e-5)


@pytest.mark.gpu
def test_pipeline_keeps_everything_on_the_device(binding):
    """render -> normalise -> network -> rescale over a batch of probes with the images left in HBM: the rendered
    images equal the host-copied ones, the network's fp32 output on the GPU agrees with the same module on the CPU,
    and every predicted hemisphere has the channel means of its rendered probe (transformMapsUpstream)."""
    torch.cuda.init()
    scene = binding.HostScene(xres=64, yres=64, spp=1)
    gpu = binding.GpuScene(scene)
    rng = np.random.default_rng(5)
    pos = rng.uniform((-150, -100, -130), (250, 150, 0), (24, 3)).astype(np.float32)
    d = rng.standard_normal((24, 3)).astype(np.float32)
    torch.manual_seed(3)
    net = ref_mod.IISPTNet()
    pipe = nn_mod.IisptPipeline(gpu, net=net, binding=binding)
    assert pipe.hip_net is not None and not hasattr(pipe, "net")   # the product holds ONE backend: the HIP kernels
    pred, inten, nrm, dist = pipe(pos, d, batch=10)
    hi, hn, hd, _ = gpu.render_probes(pos, d)
    assert np.array_equal(inten.cpu().numpy(), hi) and np.array_equal(nrm.cpu().numpy(), hn) and np.array_equal(dist.cpu().numpy(), hd)
    x, means = ref_mod.normalize_downstream(torch.from_numpy(hi), torch.from_numpy(hn), torch.from_numpy(hd))
    with torch.no_grad():
        want = ref_mod.transform_upstream(net.cpu().eval()(x), means).numpy()
    got = pred.cpu().numpy()
    assert np.isfinite(got).all()
    # measured bounds (the MIOpen era's 2e-3 / 1e-3 are gone): per element inside north_star's band, 3e-6 of the maximum at worst
    _, _, worst = assert_per_element(got, want, "pipeline (render -> normalise -> network -> rescale), 24 probes")
    assert worst < 1e-5
    got_means = got.reshape(24, -1, 3).astype(np.float64).mean(1)
    lit = (means.numpy() > 1e-6) & (got_means > 0)   # a channel the (random) network leaves at 0 stays 0: mul = 0
    assert lit.sum() >= 8
    assert np.allclose(got_means[lit], means.numpy()[lit], rtol=1e-5)


@pytest.mark.gpu
def test_indirect_film_of_a_small_frame_against_the_fp32_module(binding):
    """What the network's arithmetic does to the FILM ("no test states what the split does to the indirect film
    after exp(v) - 1 and the rescale"): the indirect pass of a 96 x 80 frame of killeroo-simple (two sweeps, radius 4 then 3; every
    stage but the network identical — same hemi points, same probe images, same gather) with the HIP network and with the fp32
    module on the CPU behind the tensor-expression transforms. Per film pixel and channel: inside north_star's 1e-4 relative band
    on >= 99.9 %, mean relative error <= 1e-5; the weights (which samples were recorded) equal."""
    torch.cuda.init()
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    scene = binding.HostScene(xres=96, yres=80, spp=1)
    gpu = binding.GpuScene(scene)
    net, _ = _recipe_net()
    n_tasks = 6 + 12
    hip = frame_mod.IisptFrame(binding, gpu, nn_mod.IisptPipeline(gpu, net=net, binding=binding))
    hip.run_batched(n_tasks, radius_start=4.0)
    ref = frame_mod.IisptFrame(binding, gpu, ref_mod.TorchPipeline(gpu, net=net, net_device="cpu"))
    ref.run_batched(n_tasks, radius_start=4.0)
    assert hip.stats == ref.stats and hip.stats["probes"] > 300
    assert torch.equal(hip.film[..., 3], ref.film[..., 3]) and float((hip.film[..., 3] > 0).double().mean()) > 0.9
    want = ref.film[..., :3].cpu().numpy()
    assert float(np.abs(want).max()) > 0
    assert_per_element(hip.film[..., :3].cpu().numpy(), want, "indirect film monitor, 96 x 80, two sweeps, HIP network vs the fp32 module")
    assert_per_element(hip.indirect_image().cpu().numpy(), ref.indirect_image().cpu().numpy(), "indirect image (normalised monitor)")
