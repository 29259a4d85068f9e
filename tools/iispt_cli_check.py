"""BASELINE config 5 through the C++ host: `iile_pbrt --integrator iispt` at 1080p (radius 10: 220 tasks, 16 direct passes) against the
Python frame (pbrt-v3-iile_amd/iispt_frame.py) — the three images bit for bit, and the command's wall time (scene load, BVH build and
image writing included). usage: python tools/iispt_cli_check.py [out.json]"""
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

b = ge._load_binding()
sys.path.insert(0, os.path.join(REPO, "tests"))
import iispt_torch_reference as ref_mod  # noqa: E402  (tests/: the PyTorch module)
nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
W, H, RADIUS, DIRECT = 1920, 1080, 10, 16
n_tasks = -(-W // (10 * RADIUS)) * -(-H // (10 * RADIUS))
torch.manual_seed(0)
module = ref_mod.IISPTNet().eval()
exe = os.path.join(REPO, "pbrt-v3-iile_amd", "lib", "iile_pbrt")
scene_file = os.path.join(REPO, "scenes", "killeroo-simple.pbrt")


def read_pfm(path):
    raw = open(path, "rb").read()
    head = f"PF\n{W} {H}\n-1.0\n".encode()
    assert raw.startswith(head)
    return np.frombuffer(raw[len(head):], "<f4").reshape(H, W, 3)[::-1]


with tempfile.TemporaryDirectory() as td:
    net = os.path.join(td, "net.iilenet")
    b.save_net_weights(module.state_dict(), net, bn_eps=module.encoder1[3].eps)
    outs = [os.path.join(td, n) for n in ("frame.pfm", "indirect.pfm", "direct.pfm")]
    env = dict(os.environ, IISPT_SCHEDULE_RADIUS_START=str(RADIUS), IILE_TIMING="1")   # (the CLI prints where its wall time went)
    cmd = [exe, scene_file, "--xres", str(W), "--yres", str(H), "--spp", "1", "--integrator", "iispt", f"--iisptNet={net}", f"--iileIndirect={n_tasks}",
           f"--iileDirect={DIRECT}", "--outfile", outs[0], f"--iisptIndirectOut={outs[1]}", f"--iisptDirectOut={outs[2]}"]
    walls, phases = [], []
    for _ in range(12):   # (every run quoted: round 5's three runs held one of 3.4 s among two of 0.65-0.70 s)
        t0 = time.time()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
        walls.append(time.time() - t0)
        assert p.returncode == 0, p.stdout
        phases.append([ln.split("timing:", 1)[1].strip() for ln in p.stdout.splitlines() if "iile_pbrt timing:" in ln or "iile timing:" in ln])
    scene = b.HostScene(xres=W, yres=H, spp=1)
    gpu = b.GpuScene(scene)
    frame = frame_mod.IisptFrame(b, gpu, nn_mod.IisptPipeline(gpu, net=module))
    frame.run_batched(n_tasks, radius_start=float(RADIUS))
    frame.run_direct(DIRECT)
    torch.cuda.synchronize()
    equal = {}
    for path, img, name in zip(outs, (frame.image(), frame.indirect_image(), frame.direct_image()), ("merged", "indirect", "direct")):
        equal[name] = bool(np.array_equal(read_pfm(path).view(np.uint32), img.cpu().numpy().view(np.uint32)))
    res = {"command": " ".join(os.path.basename(c) if os.sep in c else c for c in cmd[:2]) + " --integrator iispt ... (1920x1080, radius 10, 16 direct passes)",
           "cli_says": [ln for ln in p.stdout.strip().splitlines() if ln.startswith("IISPT:")][-1], "wall_seconds_every_run": [round(w, 3) for w in walls],
           "phases_of_the_slowest_run": phases[int(np.argmax(walls))], "phases_of_the_fastest_run": phases[int(np.argmin(walls))],
           "wall_note": "whole process: HIP start-up, scene parse + BVH build on the host, upload, the frame, three 1080p PFM images written",
           "images_equal_to_the_python_frame_bit_for_bit": equal, "python_frame_stats": frame.stats}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)
    assert all(equal.values())
