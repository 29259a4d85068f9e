"""The PyTorch statement of the IISPT network stage — TEST infrastructure, not product (VERDICT r05 weak #11 moved it here from
pbrt-v3-iile_amd/iispt_nn.py): the module and the two transforms the HIP kernels of csrc/device/iispt_net.hip are held against.

* `IISPTNet`: the U-Net of `ml/iispt_net.py:8-109` (K = 64; 7 -> 3 channels at 32 x 32). Its `state_dict` has the
  reference's parameter names and shapes, so a checkpoint trained with the reference's `ml/main_train.py` loads
  unchanged. **No weights ship with the reference**: with random weights the output means nothing; what is pinned is the
  function — `tests/golden/iispt_net_fixture.npz` holds a forward of the REFERENCE's module (imported in the build
  container, weights from the recipe of tests/iispt_net_recipe.py) and the pipe's wire order as `read_input` /
  `output_to_stdout` of `ml/main_stdio_net.py:47-86` produce it; this module must reproduce both (tests/test_iispt_nn.py).
* `normalize_downstream` / `transform_upstream`: normalizeMapsDownstream / transformMapsUpstream
  (`src/integrators/iisptrenderrunner.cpp:1041-1133`) per probe, in the reference's arithmetic (double sums for the means,
  `log(1.0 + v)` / `exp(v) - 1.0` in double, everything else in float).
* `TorchPipeline`: probe pass (HIP) -> these tensor expressions -> the eager PyTorch module (MIOpen convolutions) — the A/B leg of
  tools/probe_bench.py and the end-to-end comparison of the tests. Only tests/, tools/ and bench.py's checking legs import this file.

Image layout: the reference's `ImageFilm` stores raster row y at index h - 1 - y (`src/film/imagefilm.cpp:26-31`,
`src/core/film.cpp:245-254`) and the network was trained on that; `iile_render_probes` returns raster order.
`normalize_downstream` flips on the way in, `transform_upstream` flips back.
"""
import torch
from torch import nn

HEMI = 32
K = 64

# One row per block: (name, layers). Layers: ("pool",), ("conv", cin, cout, kernel), ("deconv", cin, cout, kernel),
# ("lrelu",), ("bn", channels), ("up",), ("relu",). Order and indices inside a block fix the state_dict keys.
_BLOCKS = (
    ("encoder0", (("conv", 7, K, 3), ("lrelu",), ("conv", K, K, 3), ("lrelu",))),
    ("encoder1", (("pool",), ("conv", K, 2 * K, 3), ("lrelu",), ("bn", 2 * K), ("conv", 2 * K, 2 * K, 3), ("lrelu",))),
    ("encoder2", (("pool",), ("conv", 2 * K, 4 * K, 3), ("lrelu",), ("bn", 4 * K), ("conv", 4 * K, 4 * K, 3), ("lrelu",))),
    ("encoder3", (("pool",), ("conv", 4 * K, 8 * K, 3), ("lrelu",), ("bn", 8 * K), ("conv", 8 * K, 4 * K, 3), ("lrelu",), ("up",))),
    ("decoder0", (("deconv", 8 * K, 4 * K, 3), ("lrelu",), ("bn", 4 * K), ("deconv", 4 * K, 2 * K, 3), ("lrelu",), ("up",))),
    ("decoder1", (("deconv", 4 * K, 2 * K, 3), ("lrelu",), ("bn", 2 * K), ("deconv", 2 * K, K, 3), ("lrelu",), ("up",))),
    ("decoder2", (("deconv", 2 * K, K, 3), ("lrelu",), ("deconv", K, K, 3), ("lrelu",), ("conv", K, 3, 1), ("relu",))),
)


def _layer(spec):
    kind = spec[0]
    if kind == "conv":
        return nn.Conv2d(spec[1], spec[2], spec[3], stride=1, padding=spec[3] // 2)
    if kind == "deconv":
        return nn.ConvTranspose2d(spec[1], spec[2], spec[3], stride=1, padding=spec[3] // 2)
    if kind == "lrelu":
        return nn.LeakyReLU(0.2)
    if kind == "relu":
        return nn.ReLU()
    if kind == "bn":
        return nn.BatchNorm2d(spec[1])
    if kind == "pool":
        return nn.MaxPool2d(2)
    if kind == "up":
        return nn.Upsample(scale_factor=2, mode="bilinear")
    raise ValueError(kind)


class IISPTNet(nn.Module):
    """ml/iispt_net.py:8-109: three encoder levels, a bottleneck that upsamples back, three decoder levels fed by
    the concatenation of the level below with the matching encoder output."""

    def __init__(self):
        super().__init__()
        for name, layers in _BLOCKS:
            setattr(self, name, nn.Sequential(*[_layer(s) for s in layers]))

    def forward(self, x):
        e0 = self.encoder0(x)
        e1 = self.encoder1(e0)
        e2 = self.encoder2(e1)
        y = self.encoder3(e2)
        y = self.decoder0(torch.cat((y, e2), 1))
        y = self.decoder1(torch.cat((y, e1), 1))
        return self.decoder2(torch.cat((y, e0), 1))


def wire_to_network_input(intensity, normals, distance):
    """The pipe's layout -> the network's (`read_input`, ml/main_stdio_net.py:47-72): three images as the runner writes them,
    (n, h, w, 3), (n, h, w, 3) and (n, h, w) in ImageFilm row order, become (n, 7, h, w): intensity RGB, normal XYZ,
    distance. Pinned against the reference's own function by tests/golden/iispt_net_fixture.npz."""
    x7 = torch.cat((intensity, normals, distance.unsqueeze(-1)), -1)
    return x7.permute(0, 3, 1, 2).contiguous()


def network_output_to_wire(out):
    """`output_to_stdout` (ml/main_stdio_net.py:77-86): the network's (n, 3, h, w) goes back as (n, h, w, 3)."""
    return out.permute(0, 2, 3, 1)


def normalize_downstream(intensity, normals, distance):
    """normalizeMapsDownstream, batched: intensity (n, h, h, 3), normals (n, h, h, 3), distance (n, h, h) in raster
    order -> network input (n, 7, h, h) float32 in the reference's row order, and the per-probe channel means
    (n, 3) that transform_upstream needs."""
    n = intensity.shape[0]
    i64 = intensity.double()
    # computeMeanChannels / computeMean: double sums over float texels (imagefilm.cpp:203-254)
    chan_mean = i64.reshape(n, -1, 3).mean(1).float()
    mean = i64.reshape(n, -1).mean(1).float()
    ratio = torch.where(mean == 0, torch.zeros_like(mean, dtype=torch.float64), 1.0 / (10.0 * mean.double())).float()
    x = intensity * ratio.view(n, 1, 1, 1)                                  # multiply(float)
    x = torch.log(1.0 + torch.clamp(x, min=0).double()).float()             # positiveLog: log(1.0 + v) in double
    x = x + torch.tensor(-0.1, dtype=torch.float32, device=x.device)        # add(-0.1)
    nrm = torch.clamp((normals - 0.0) / 1.0, -1.0, 1.0)                      # normalize(-1, 1): mid 0, r 1
    z_mean = distance.double().reshape(n, -1).mean(1).float()
    d = distance + 1.0
    div = (10.0 * (z_mean.double() + 1.0)).float()
    div = torch.where(div == 0, torch.ones_like(div), div)
    d = d * (1.0 / div.double()).float().view(n, 1, 1)
    d = torch.log(1.0 + torch.clamp(d, min=0).double()).float()
    d = d + torch.tensor(-0.1, dtype=torch.float32, device=d.device)
    # ImageFilm row = h - 1 - y; then the pipe's (h, w, c) images become (channels, height, width) per probe
    return wire_to_network_input(torch.flip(x, dims=(1,)), torch.flip(nrm, dims=(1,)), torch.flip(d, dims=(1,))), chan_mean


def transform_upstream(out, chan_mean):
    """transformMapsUpstream, batched: network output (n, 3, h, h) -> predicted intensity (n, h, h, 3) in raster order,
    rescaled so that each channel's mean is the rendered probe's."""
    n = out.shape[0]
    y = torch.exp(torch.clamp(out.float(), min=0).double()) - 1.0           # positiveLogInverse in double
    y = y.float()
    actual = y.double().reshape(n, 3, -1).mean(2).float()                   # computeMeanChannels
    mul = torch.where(actual > 1e-10, chan_mean / actual, torch.zeros_like(actual))
    y = y * mul.view(n, 3, 1, 1)
    return torch.flip(network_output_to_wire(y), dims=(1,)).contiguous()


class TorchPipeline:
    """render probes (HIP, through the C ABI) -> normalise -> the eager PyTorch module -> rescale: the same call signature as
    pbrt-v3-iile_amd/iispt_nn.IisptPipeline, for A/B timing and end-to-end comparisons only."""

    def __init__(self, gpu_scene, net=None, dtype=torch.float32, device="cuda", net_device=None):
        """net_device="cpu": the module itself runs on the CPU (the reference's own arithmetic: fp32, no MIOpen) and only the probe
        pass is the GPU's — the statement the HIP network's frames are compared with."""
        self.gpu = gpu_scene
        self.device = torch.device(device)
        self.net_device = torch.device(net_device) if net_device is not None else self.device
        self.dtype = dtype
        self.net = (net if net is not None else IISPTNet()).eval().to(self.net_device)
        if dtype != torch.float32:
            self.net = self.net.to(dtype)
        if self.net_device.type == "cuda":
            self.net = self.net.to(memory_format=torch.channels_last)
        self.events = None

    def _timed(self, stage, fn):
        if self.events is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.events.append((stage, e0, e1))
        return out

    def infer(self, x):
        if self.net_device.type != "cuda":
            return self.net(x.to(self.net_device, self.dtype)).float().to(self.device)
        return self.net(x.to(self.dtype).contiguous(memory_format=torch.channels_last)).float()

    @torch.no_grad()
    def __call__(self, pos, direction, batch=32768, film_rows=False, pred_out=None, slot=None):
        n = len(pos)
        inten = torch.empty((n, HEMI, HEMI, 3), dtype=torch.float32, device=self.device)
        nrm = torch.empty((n, HEMI, HEMI, 3), dtype=torch.float32, device=self.device)
        dist = torch.empty((n, HEMI, HEMI), dtype=torch.float32, device=self.device)
        self._timed("probe_pass", lambda: self.gpu.render_probes(pos, direction, device_out=(inten.data_ptr(), nrm.data_ptr(), dist.data_ptr())))
        pred = torch.empty_like(inten)
        for first in range(0, n, batch):
            sl = slice(first, min(n, first + batch))
            if self.net_device.type != "cuda":   # the whole statement on the CPU
                x, means = normalize_downstream(inten[sl].cpu(), nrm[sl].cpu(), dist[sl].cpu())
                pred[sl] = transform_upstream(self.net(x), means).to(self.device)
                continue
            x, means = self._timed("normalize", lambda: normalize_downstream(inten[sl], nrm[sl], dist[sl]))
            y = self._timed("network", lambda: self.infer(x))
            pred[sl] = self._timed("rescale", lambda: transform_upstream(y, means))
        if film_rows:
            pred = torch.flip(pred, dims=(1,))
        if pred_out is not None:
            pred_out[slot.long()] = pred
            pred = pred_out
        return pred, inten, nrm, dist
