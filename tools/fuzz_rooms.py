"""One-off robustness sweep: differently seeded box rooms (tests/boxroom.py) under every light set-up and material mix,
odd resolutions, sample counts and depths — the uninstrumented and the instrumented film against the oracle, counters too.
With `iispt` as a third argument the IISPT stages run on every room as well: two direct passes, the runner's hemi points and its
gather over random hemispheres, each against the oracle bit for bit.
usage: python tools/fuzz_rooms.py [first_seed=100] [n=24] [iispt]"""
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
iispt = len(sys.argv) > 3 and sys.argv[3] == "iispt"
lights = ["area", "quad", "multi", "spot", "point", "envmap", "sky"]  # (the last two: infinite lights — k_mis ends at the first hit)
mats = ["plain", "all", "mixed", "ubertrans", "roughglass", "aniso"]   # (round 6: uber with opacity < 1 and Kt; rough glass; uroughness != vroughness)
bad = 0
with tempfile.TemporaryDirectory() as td:
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        kw = dict(xres=int(rng.integers(17, 120)), yres=int(rng.integers(9, 90)), spp=int(rng.integers(1, 6)), ico_levels=int(rng.integers(1, 4)),
                  n_blobs=int(rng.integers(1, 12)), wall_n=int(rng.integers(2, 16)), seed=seed, maxdepth=int(rng.integers(1, 8)),
                  light=lights[seed % len(lights)], materials=mats[(seed // len(lights)) % len(mats)])
        path = os.path.join(td, "room.pbrt")
        if kw["light"] == "envmap":  # the map's directory; every other seed with image textures / bump maps / alpha masks as well
            if seed % 2:
                kw["textures"] = os.path.join(td, f"tex{seed}")
            else:
                kw["env_dir"] = os.path.join(td, f"env{seed}")
        try:
            text = boxroom.boxroom_pbrt(**kw)
            if seed % 5 == 0:   # "pixelbounds" (round 6): a random rectangle x0 x1 y0 y1, sometimes reaching past the film
                xs = sorted(int(v) for v in rng.integers(-8, kw["xres"] + 8, 2))
                ys = sorted(int(v) for v in rng.integers(-8, kw["yres"] + 8, 2))
                text = text.replace('Integrator "path"', 'Integrator "path" "integer pixelbounds" [%d %d %d %d]' % (xs[0], xs[1] + 1, ys[0], ys[1] + 1))
                kw["materials"] += "+pb"
            open(path, "w").write(text)
        except TypeError as e:
            print("skip", kw, e)
            continue
        try:
            scene = b.HostScene(path=path)
        except RuntimeError as e:
            print("seed", seed, "not loadable:", str(e)[:100])
            continue
        gpu = b.GpuScene(scene)
        ref, ost = o.render(scene)
        film, st = gpu.render(collect_stats=True)
        plain, pst = gpu.render(spp_per_pass=int(rng.integers(0, 3)))
        ok = np.array_equal(film.view(np.uint32), ref.view(np.uint32)) and np.array_equal(plain.view(np.uint32), ref.view(np.uint32))
        ok = ok and st["closest_rays"] == ost["regular_rays"] and st["shadow_rays"] == ost["shadow_rays"] and st["nodes_closest"] == ost["nodes_closest"]
        stages = ""
        if iispt:
            try:
                direct = gpu.render_direct(2)
                ok_d = np.array_equal(direct.view(np.uint64), o.iispt_direct(scene, 2).view(np.uint64))
                task = b.IisptTask(0, 0, kw["xres"], kw["yres"], int(rng.integers(3, 12)), 0, 0)
                valid, pos, dr = gpu.iispt_hemi_points(task)
                rv, rp, rd = o.iispt_hemi_points(scene, task)
                ok_h = np.array_equal(valid, rv) and np.array_equal(pos.view(np.uint32), rp.view(np.uint32)) and np.array_equal(dr.view(np.uint32), rd.view(np.uint32))
                nn = rng.uniform(0.0, 3.0, valid.shape + (32, 32, 3)).astype(np.float32)
                out = gpu.iispt_gather(task, valid, pos, dr, nn)
                ok_g = np.array_equal(out.view(np.uint32), o.iispt_gather(scene, task, valid, pos, dr, nn).view(np.uint32))
                stages = f" iispt direct/hemi/gather {'OK' if ok_d else 'MISMATCH'}/{'OK' if ok_h else 'MISMATCH'}/{'OK' if ok_g else 'MISMATCH'}"
                ok = ok and ok_d and ok_h and ok_g
            except RuntimeError as e:
                stages = " iispt refused: " + str(e)[:80]
        print("seed", seed, kw["light"], kw["materials"], f'{kw["xres"]}x{kw["yres"]}x{kw["spp"]} depth {kw["maxdepth"]}', "OK" if ok else "MISMATCH",
              "traced", pst["ext_rays_traced"], "of", st["ext_rays"], stages)
        bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
