"""The network stage behind the IISPT probe pass, in-process on the GPU.

The reference pipes every probe through a child Python process (`ml/main_stdio_net.py`, 47 ms per probe,
`Doc.md:58-61`): `IisptRenderRunner` normalises the three probe images (`normalizeMapsDownstream`,
`src/integrators/iisptrenderrunner.cpp:1041-1092`), writes 32*32*7 floats to the pipe, reads 32*32*3 back and
rescales them (`transformMapsUpstream`, `:1095-1133`). Here the probe images never leave HBM: `iile_render_probes`
(include/iile_gpu.h) writes them into device tensors and `iile_iispt_net_predict` runs the two transforms and `IISPTNet.forward`
(`ml/iispt_net.py:8-109`) over the whole batch on the hand-written kernels of csrc/device/iispt_net.hip.

This file is plumbing over the C ABI and holds ONE backend: no PyTorch module, no eager fallback. Weights arrive as a `state_dict`
with the reference's parameter names and shapes (what `torch.load` of a checkpoint of the reference's `ml/main_train.py` yields;
binding.GpuNet packs it). The PyTorch statement of the network and of the transforms that the kernels are tested against lives in
tests/iispt_torch_reference.py.

Image layout: the reference's `ImageFilm` stores raster row y at index h - 1 - y (`src/film/imagefilm.cpp:26-31`,
`src/core/film.cpp:245-254`) and the network was trained on that; `iile_render_probes` returns raster order and
`iile_iispt_net_predict` flips on the way in and back on the way out (film_rows: leaves the network's own row order, which is what
`iile_iispt_gather` reads).
"""
import torch

HEMI = 32
BN_EPS = 1e-5   # nn.BatchNorm2d's default, which ml/iispt_net.py:27-88 does not override


def _load_binding():
    """binding.py beside this file, under the name __graft_entry__ / tests/conftest.py load it by (one copy per process)."""
    import importlib.util
    import os
    import sys
    if "iile_binding" in sys.modules:
        return sys.modules["iile_binding"]
    spec = importlib.util.spec_from_file_location("iile_binding", os.path.join(os.path.dirname(os.path.abspath(__file__)), "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["iile_binding"] = mod
    spec.loader.exec_module(mod)
    return mod


class IisptPipeline:
    """render probes -> normalise -> network -> rescale, everything resident in HBM, through the C ABI
    (iile_render_probes, iile_iispt_net_predict; binding.GpuNet). `net` supplies the weights: a state_dict (name -> tensor / array,
    the reference's names) or anything with a `.state_dict()` (e.g. a module with a checkpoint of the reference's training loaded).
    Without the HIP library the constructor raises: there is no other backend."""

    def __init__(self, gpu_scene, net, device="cuda", binding=None, bn_eps=BN_EPS, batch=8192):
        """batch: probes per set of network launches (the activation workspace is 1.19 MiB x batch; 8 192 is as fast as any larger one;
        iile_iispt_net_predict cuts a larger call into equal sets and halves the batch if the device is short of memory)."""
        self.gpu = gpu_scene
        self.batch = int(batch)
        self.device = torch.device(device)
        if binding is None:
            binding = _load_binding()
        state = net.state_dict() if hasattr(net, "state_dict") else net
        self.hip_net = binding.GpuNet(state, bn_eps=bn_eps)
        self.events = None   # bench.py: a list that receives (stage, start event, end event) on the current stream

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
        """(n, 7, h, h) float32 on the device -> (n, 3, h, h) float32: the bare network (iile_iispt_net_forward)."""
        x = x.contiguous()
        y = torch.empty((x.shape[0], 3, HEMI, HEMI), dtype=torch.float32, device=x.device)
        self.hip_net.forward(x.data_ptr(), y.data_ptr(), x.shape[0], stream=torch.cuda.current_stream().cuda_stream)
        return y

    @torch.no_grad()
    def __call__(self, pos, direction, batch=None, film_rows=False, pred_out=None, slot=None):
        """(n, 3) probe origins and directions -> (predicted intensity (n, h, h, 3), rendered intensity, normals,
        distance), all torch tensors on the device, raster order (film_rows: the prediction in the network's own row order,
        ImageFilm's, as iile_iispt_gather reads it). The two transforms run inside iile_iispt_net_predict.
        pred_out (m, h, h, 3) with slot (n,) int32 on the device: probe i's prediction is written to pred_out[slot[i]] (the frame keeps
        one image per hemi point, valid or not) and pred_out is returned in place of the (n, ...) tensor.
        batch: this call's probes per set of network launches (default: the pipeline's)."""
        n = len(pos)
        batch = self.batch if batch is None else int(batch)
        inten = torch.empty((n, HEMI, HEMI, 3), dtype=torch.float32, device=self.device)
        nrm = torch.empty((n, HEMI, HEMI, 3), dtype=torch.float32, device=self.device)
        dist = torch.empty((n, HEMI, HEMI), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream().cuda_stream   # every stage of the indirect pass goes on the caller's current stream
        self._timed("probe_pass", lambda: self.gpu.render_probes(pos, direction, device_out=(inten.data_ptr(), nrm.data_ptr(), dist.data_ptr()), stream=stream))
        pred = pred_out if pred_out is not None else torch.empty_like(inten)
        self._timed("network", lambda: self.hip_net.predict(inten.data_ptr(), nrm.data_ptr(), dist.data_ptr(), pred.data_ptr(), n, film_rows=film_rows,
                                                            max_batch=batch, stream=stream,
                                                            slot_ptr=slot.data_ptr() if pred_out is not None else None))
        return pred, inten, nrm, dist
