"""One-off robustness sweep of the IISPT direct pass (iile_render_direct): differently seeded box rooms (tests/boxroom.py) under
every light set-up — infinite lights included — and material mix (mirrors, uber, glass: the per-pixel recursion tree), with and
without image textures, odd resolutions — the film monitor's doubles against the oracle's restatement of
DirectProgressiveIntegrator, bit for bit; scenes the pass refuses (textures on a specular sphere) must be refused by both.
usage: python tools/fuzz_direct.py [first_seed=100] [n=24]"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as ge  # noqa: E402
import boxroom  # noqa: E402
import oracle_binding  # noqa: E402

b = ge._load_binding()
o = oracle_binding.Oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
lights = ["area", "quad", "multi", "spot", "point", "envmap", "sky", "many"]
mats = ["plain", "all", "mixed", "ubertrans", "roughglass", "aniso"]   # (the last three: round 6)
bad = 0
with tempfile.TemporaryDirectory() as td:
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        kw = dict(xres=int(rng.integers(17, 100)), yres=int(rng.integers(9, 70)), spp=1, ico_levels=int(rng.integers(1, 4)),
                  n_blobs=int(rng.integers(1, 10)), wall_n=int(rng.integers(2, 12)), seed=seed, light=lights[seed % len(lights)],
                  materials=mats[(seed // len(lights)) % len(mats)])
        if kw["light"] == "envmap" or seed % 3 == 0:
            if seed % 2:
                kw["textures"] = os.path.join(td, f"tex{seed}")
            elif kw["light"] == "envmap":
                kw["env_dir"] = os.path.join(td, f"env{seed}")
        path = os.path.join(td, "room.pbrt")
        try:
            open(path, "w").write(boxroom.boxroom_pbrt(**kw))
            scene = b.HostScene(path=path)
        except (TypeError, RuntimeError, AssertionError) as e:
            print("seed", seed, "skipped:", str(e)[:100])
            continue
        gpu = b.GpuScene(scene)
        passes, first_pass = int(rng.integers(1, 11)), int(rng.integers(0, 5))   # (four passes run per set of launches: 1 .. 10 crosses that)
        what = f'{kw["light"]} {kw["materials"]}{" textured" if "textures" in kw else ""} {kw["xres"]}x{kw["yres"]} passes {first_pass}+{passes}'
        try:
            ref = o.iispt_direct(scene, passes, first_pass=first_pass)
        except RuntimeError as e:
            try:
                gpu.render_direct(passes, first_pass=first_pass)
                print("seed", seed, what, "MISMATCH: the oracle refuses (", str(e)[:60], ") and the device renders")
                bad += 1
            except RuntimeError:
                print("seed", seed, what, "refused by both")
            continue
        got = gpu.render_direct(passes, first_pass=first_pass)
        ok = np.array_equal(got.view(np.uint64), ref.view(np.uint64))
        print("seed", seed, what, "OK" if ok else f"MISMATCH ({int((got != ref).any(axis=2).sum())} pixels)", "mean", float(got[..., :3].sum() / max(got[..., 3].sum(), 1)))
        bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
