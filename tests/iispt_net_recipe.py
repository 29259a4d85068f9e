"""Deterministic weights for the IISPTNet parity fixture (tests/golden/iispt_net_fixture.npz).

No trained weights ship with the reference, and 5.5 M random floats are too much to commit, so the fixture stores a
RECIPE instead: every tensor of the reference's `state_dict` is filled from a counter-based generator (splitmix64 of
the element number, 24 random bits per element, float64 arithmetic rounded once to float32) that gives the same bits on
any machine. `tests/golden/make_iispt_net_fixture.py` fills the REFERENCE's module (imported from /root/reference/ml in
the build container) this way and stores the tensor names, shapes and SHA-256 sums beside the input and the output;
the tests fill `iispt_nn.IISPTNet` the same way and must reproduce that output."""
import hashlib

import numpy as np

SEED = 0x11D0_2026_1003


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)).astype(np.uint64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)).astype(np.uint64)
    return z ^ (z >> np.uint64(31))


def uniform01(stream, n):
    """n numbers in [0, 1) with 24 random bits each (exact in float32), stream-th sequence."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40)) + np.uint64(SEED)
        z = _splitmix64(i)
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def tensor_for(index, name, shape):
    """The recipe's value for state_dict entry number `index` called `name` (numpy array of the entry's dtype)."""
    n = int(np.prod(shape)) if len(shape) else 1
    if name.endswith("num_batches_tracked"):
        return np.array(7, np.int64)
    u = uniform01(index, n)
    if len(shape) == 4:                       # convolution / transposed convolution kernels: He-uniform for LeakyReLU(0.2)
        fan = n / shape[0]
        v = (2 * u - 1) * np.sqrt(6.0 / (1.04 * fan))
    elif name.endswith("running_var"):
        v = 0.5 + u
    elif name.endswith("running_mean"):
        v = (2 * u - 1) * 0.1
    elif name.endswith("weight"):             # BatchNorm scale
        v = 0.8 + 0.4 * u
    elif n == 3:                              # the output layer's bias: positive, so that the final ReLU lets most of the image through
        v = 3.0 + 1.5 * u
    else:                                     # biases
        v = (2 * u - 1) * 0.05
    return v.astype(np.float32).reshape(shape)


def fill_state_dict(module):
    """Overwrite every entry of module.state_dict() by the recipe (in state_dict order); returns [(name, shape, sha256)]."""
    import torch
    sd = module.state_dict()
    out = []
    for index, (name, t) in enumerate(sd.items()):
        a = tensor_for(index, name, tuple(t.shape))
        sd[name] = torch.from_numpy(np.ascontiguousarray(a)).to(t.dtype)
        out.append((name, tuple(int(s) for s in t.shape), hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()))
    module.load_state_dict(sd, strict=True)
    return out


def fixture_input(n=4, h=32):
    """A seeded (n, 7, h, h) network input in the ranges normalizeMapsDownstream produces: log-intensity in [-0.1, 1.2],
    normals in [-1, 1], log-distance in [-0.1, 0.6]."""
    u = uniform01(1000, n * 7 * h * h).reshape(n, 7, h, h)
    x = np.empty((n, 7, h, h), np.float64)
    x[:, 0:3] = -0.1 + 1.3 * u[:, 0:3] ** 2
    x[:, 3:6] = 2 * u[:, 3:6] - 1
    x[:, 6] = -0.1 + 0.7 * u[:, 6]
    return x.astype(np.float32)
