"""What the split-bf16 arithmetic of csrc/device/iispt_net.hip does to IISPTNet's output, emulated on the CPU (no GPU needed):
every convolution's operands cut into bf16 pieces (a = a_hi + a_lo (+ a_lo2)), the chosen products accumulated in fp32, the rest
of the module (LeakyReLU, BatchNorm, pooling, upsampling) in fp32 as the module has it. Compared per element with the fp32 module
and with the module evaluated in float64 (the value both approximate).

    python tools/net_split_emulation.py [n_probes=64]

Prints, per variant, the share of output elements inside |err| <= 1e-4 |want| + 1e-6 max|want| (VERDICT r05 "next" 2), the mean
relative error, and max|err| / max|want|. The fp32 module against float64 is the floor: what "fp32" itself means on this network."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
import iispt_net_recipe as recipe            # noqa: E402
import iispt_torch_reference as ref_mod      # noqa: E402


def split(t, pieces):
    out, rest = [], t
    for _ in range(pieces):
        p = rest.bfloat16().float()
        out.append(p)
        rest = rest - p
    return out


def split_conv(x, m, products):
    """products: list of (i, j): piece i of the activations times piece j of the weights."""
    pa = split(x, 1 + max(i for i, _ in products))
    pw = split(m.weight.detach(), 1 + max(j for _, j in products))
    f = F.conv_transpose2d if isinstance(m, torch.nn.ConvTranspose2d) else F.conv2d
    y = None
    for i, j in sorted(products, key=lambda p: -(p[0] + p[1])):   # small terms first, as the kernel orders them
        t = f(pa[i], pw[j], None, stride=1, padding=m.padding)
        y = t if y is None else y + t
    return y + m.bias.detach().view(1, -1, 1, 1)


def forward(net, x, products, first_layer_products=None):
    first = [True]

    def run(block, t):
        for m in block:
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and m.kernel_size == (3, 3) and products is not None:
                pr = first_layer_products if (first[0] and first_layer_products) else products
                first[0] = False
                t = split_conv(t, m, pr)
            else:
                t = m(t)
        return t
    e0 = run(net.encoder0, x)
    e1 = run(net.encoder1, e0)
    e2 = run(net.encoder2, e1)
    y = run(net.encoder3, e2)
    y = run(net.decoder0, torch.cat((y, e2), 1))
    y = run(net.decoder1, torch.cat((y, e1), 1))
    return run(net.decoder2, torch.cat((y, e0), 1))


def score(got, want):
    got, want = got.double().numpy().ravel(), want.double().numpy().ravel()
    mx = np.abs(want).max()
    err = np.abs(got - want)
    inside = err <= 1e-4 * np.abs(want) + 1e-6 * mx
    nz = np.abs(want) > 1e-6 * mx
    return {"inside_frac": float(inside.mean()), "mean_rel_err_where_nonzero": float((err[nz] / np.abs(want[nz])).mean()),
            "max_err_over_max": float(err.max() / mx), "p999_err_over_max": float(np.quantile(err, 0.999) / mx)}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    torch.manual_seed(0)
    net = ref_mod.IISPTNet()
    recipe.fill_state_dict(net)
    net.eval()
    x = torch.from_numpy(recipe.fixture_input(n))
    variants = {
        "bf16 plain (hh)": [(0, 0)],
        "3 products (hh, hl, lh): the shipped kernels of round 5": [(0, 0), (0, 1), (1, 0)],
        "4 products (+ ll)": [(0, 0), (0, 1), (1, 0), (1, 1)],
        "6 products, three pieces (hh, hm, mh, hl, lh, mm)": [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)],
    }
    with torch.no_grad():
        y32 = net(x)
        y64 = net.double()(x.double())
        net.float()
        res = {"fp32 module vs float64": score(y32, y64)}
        for name, pr in variants.items():
            y = forward(net, x, pr)
            res[name + " vs fp32 module"] = score(y, y32)
            res[name + " vs float64"] = score(y, y64)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
