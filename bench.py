#!/usr/bin/env python3
"""bench.py — Mray/s and samples/s of the HIP path tracer on killeroo-simple 1080p.

    python bench.py --gpus N --steps K --warmup W

A *step* is one complete pass of the hot path over the configured workload:
SamplerIntegrator::Render of killeroo-simple at 1920x1080, 64*N pixel samples,
16x16 tiles interleaved over the N ranks (one process per GPU; for N > 1 the
driver launches this file through torch.distributed.run), finished by ONE
sum-reduction of the {X,Y,Z,w} film to rank 0 over RCCL. Per-GPU work is fixed
as N grows (each rank renders 1/N of the tiles at 64*N spp), so scaling is weak.
The film stays in HBM for the whole timed region.

Rank 0 prints one JSON line. `value` is whole-job Mray/s (rays = Scene::Intersect +
Scene::IntersectP calls, the reference's own ray definition); `roofline` prices
the dominant kernel against the HBM roof using SURVEY.md §8(d)'s algorithmic
bytes per ray; `cpu_baseline` is the CPU oracle timed on this box's host cores on
a bounded sample of the same workload (baseline only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(rays, nodes, tris):
    """B = 32 B per BVH node visited + 48 B per triangle test +
    48 B of queue traffic per ray (32 B ray record in, 16 B hit record out)."""
    return 32 * nodes + 48 * tris + 48 * rays


def measured_copy_gbs(torch):
    """Device-to-device copy rate on this box (read + write bytes / time): the practical HBM
    ceiling next to the 8 TB/s spec figure (SURVEY.md 8d asks for both)."""
    n = 1 << 28  # 1 GiB of float32
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(scene_path, xres, yres, target_seconds, workload_name="killeroo-simple"):
    """Time the CPU oracle (restatement of the reference path, all host cores, 16x16
    tile self-scheduling, render loop only) on a bounded number of pixel samples."""
    import __graft_entry__ as ge
    import oracle_binding as ob
    b = ge._load_binding()
    threads = os.cpu_count() or 1
    scene = b.HostScene(path=scene_path, xres=xres, yres=yres, spp=64)
    orc = ob.Oracle()
    _, st = orc.render(scene, trig_mode=ob.TRIG_LIBM, threads=threads, k_begin=0, k_end=1)
    t1 = max(st["seconds"], 1e-3)
    n = int(max(1, min(16, round(target_seconds / t1))))
    _, st = orc.render(scene, trig_mode=ob.TRIG_LIBM, threads=threads, k_begin=1, k_end=1 + n)
    rays = st["regular_rays"] + st["shadow_rays"]
    return {
        "value": round(rays / st["seconds"] / 1e6, 3),
        "unit": "Mray/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{workload_name} {xres}x{yres}, pixel samples k=1..{n} of 64 ({st['camera_rays']} camera samples, "
                  f"{rays} rays, {st['seconds']:.2f} s, libm trig)",
        "msamples_per_s": round(st["camera_rays"] / st["seconds"] / 1e6, 4),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64, help="pixel samples per GPU-equivalent (total = spp * gpus)")
    ap.add_argument("--spp-per-pass", type=int, default=0)
    ap.add_argument("--scene", default=os.path.join(REPO, "scenes", "killeroo-simple.pbrt"))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the cpu_baseline leg")
    ap.add_argument("--workload", choices=["killeroo", "boxroom", "boxroom-textured"], default="killeroo",
                    help="boxroom: the synthetic ~287 k-triangle closed room of tests/boxroom.py (deep-BVH stress, "
                         "SURVEY.md 8d's stand-in for the Sponza config that does not ship with the reference); "
                         "boxroom-textured: the same room open to an environment-mapped sky, with image textures, "
                         "alpha masks and specular materials (the whole feature set of SURVEY.md 8 f1)")
    args = ap.parse_args()
    workload_name = "killeroo-simple"
    if args.workload in ("boxroom", "boxroom-textured"):
        import tempfile
        import boxroom
        tmp = tempfile.NamedTemporaryFile("w", suffix=".pbrt", delete=False)
        if args.workload == "boxroom":
            tmp.write(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64))
            workload_name = "synthetic boxroom (287k triangles, tests/boxroom.py)"
        else:
            tmp.write(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64, light="envmap", materials="mixed",
                                           textures=tempfile.mkdtemp(prefix="boxroom_img_")))
            workload_name = "synthetic textured boxroom (266k triangles, environment map, image textures, alpha masks; tests/boxroom.py)"
        tmp.close()
        args.scene = tmp.name

    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.build_if_needed()
    b = ge._load_binding()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    total_spp = args.spp * world
    scene = b.HostScene(path=args.scene, xres=args.xres, yres=args.yres, spp=total_spp)
    gpu = b.GpuScene(scene)
    h, w = scene.film_shape
    film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    import importlib.util
    spec = importlib.util.spec_from_file_location("iile_multigpu", os.path.join(REPO, "pbrt-v3-iile_amd", "multigpu.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)

    def step(collect=False, timed=False):
        box = {}

        def render(tile_rank, tile_nranks):
            _, box["st"] = gpu.render(tile_rank=tile_rank, tile_nranks=tile_nranks, spp_per_pass=args.spp_per_pass,
                                      collect_stats=collect, time_kernels=timed, film_device_ptr=film.data_ptr(),
                                      stream=stream, want_stats=True)

        mg.render_sharded(render, film, dist)  # N > 1: one sum-reduction of the film shards to rank 0 (RCCL)
        return box["st"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # instrumented pass (untimed): ray / node / triangle counts of one step on this rank
    cst = step(collect=True)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    agg = {"ms_extend": 0.0, "ms_connect": 0.0, "ms_shadow": 0.0, "ms_mis": 0.0, "ms_resolve": 0.0, "ms_shade": 0.0, "ms_generate": 0.0,
           "ms_film": 0.0, "ms_total": 0.0, "n_extend_launches": 0, "n_connect_launches": 0}
    for _ in range(args.steps):
        st = step(timed=True)
        for k in agg:
            agg[k] += st[k]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([cst["closest_rays"], cst["shadow_rays"], cst["camera_rays"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        rays_closest, rays_shadow, cam = (int(x) for x in cnt.tolist())
    else:
        rays_closest, rays_shadow, cam = cst["closest_rays"], cst["shadow_rays"], cst["camera_rays"]

    if rank == 0:
        rays_step = rays_closest + rays_shadow
        ms_per_step = elapsed * 1e3 / args.steps
        mray = rays_step * args.steps / elapsed / 1e6
        # per-ray averages of this rank's instrumented step
        r_all = cst["closest_rays"] + cst["shadow_rays"]
        n_node = (cst["nodes_closest"] + cst["nodes_any"]) / max(r_all, 1)
        n_tri = cst["tri_tests"] / max(r_all, 1)
        b_ray = 32 * n_node + 48 * n_tri + 48
        # dominant kernel by measured HIP-event time on this rank
        ext_bytes = algorithmic_bytes(cst["ext_rays"], cst["ext_nodes"], cst["ext_tri_tests"])
        sh_bytes = algorithmic_bytes(cst["shadow_rays"], cst["nodes_any"], cst["any_tri_tests"])
        mis_rays = cst["closest_rays"] - cst["ext_rays"]
        mis_bytes = algorithmic_bytes(mis_rays, cst["nodes_closest"] - cst["ext_nodes"],
                                      cst["tri_tests"] - cst["ext_tri_tests"] - cst["any_tri_tests"])
        kernels = {
            "k_extend": (agg["ms_extend"], agg["n_extend_launches"], ext_bytes, cst["ext_rays"]),
            "k_shadow": (agg["ms_shadow"], agg["n_connect_launches"], sh_bytes, cst["shadow_rays"]),
            "k_mis": (agg["ms_mis"], agg["n_connect_launches"], mis_bytes, mis_rays),
        }
        dom = max(kernels, key=lambda k: kernels[k][0])
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3
        # runs of this same command: tools/collect_profiles.sh -> profiles/*_pmc_traffic.json)
        traffic, traffic_src = None, None
        import glob
        for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc_traffic.json")))[::-1]:
            try:
                tj = json.load(open(f))
                # the product builds of the kernel (no counting, no alpha; k_extend has one more template argument:
                # its first launch of a step also makes the camera rays), averaged over the launches of a step
                ents = [v for k, v in tj["kernels"].items() if k.startswith(dom + "<false, false") or k == dom + "<false>"]
                if ents and world == 1 and (args.xres, args.yres, args.spp, args.workload) == (1920, 1080, 64, "killeroo"):
                    n_l = sum(e.get("launches_in_step", 1) for e in ents)
                    traffic = int(sum(e["hbm_bytes_per_launch"] * e.get("launches_in_step", 1) for e in ents) / n_l)
                    traffic_src = os.path.basename(f)
                    break
            except Exception:
                pass
        # every pipeline kernel priced the same way (HIP-event time of this rank, algorithmic bytes)
        nee = cst["nee_evals"]
        next_rays = cst["ext_rays"] - cst["camera_rays"]
        other = {
            "k_shade": (agg["ms_shade"], 48 * nee + 112 * nee + 32 * next_rays,
                        "approx.: 48 B in per hit, 112 B NEE record + 32 B next ray out; VALU bound"),
            "k_mis_lit": (agg["ms_resolve"], 1 * nee, "one result byte per NEE record"),
            "k_generate": (agg["ms_generate"], 52 * cst["camera_rays"], "ray, Halton index, L written"),
            "k_film": (agg["ms_film"], 16 * cst["camera_rays"], "L read per sample"),
        }
        per_kernel = {}
        for k, (ms_k_, nl_, by_, _r) in kernels.items():
            per_kernel[k] = {"ms_per_step": round(ms_k_ / args.steps, 3),
                             "algorithmic_gbs": round(by_ * args.steps / max(ms_k_, 1e-9) / 1e6, 1)}
        for k, (ms_k_, by_, note_) in other.items():
            if ms_k_ <= 0:  # k_generate when camera rays are made inside k_extend's first launch: nothing to price
                per_kernel[k] = {"ms_per_step": 0.0, "algorithmic_gbs": None, "note": note_ + " (not launched: fused into k_extend)"}
                continue
            per_kernel[k] = {"ms_per_step": round(ms_k_ / args.steps, 3),
                             "algorithmic_gbs": round(by_ * args.steps / max(ms_k_, 1e-9) / 1e6, 1), "note": note_}
        for k in per_kernel:
            g_ = per_kernel[k]["algorithmic_gbs"]
            per_kernel[k]["frac_of_hbm_peak"] = None if g_ is None else round(g_ / HBM_PEAK_GBS, 4)
        copy_gbs = measured_copy_gbs(torch)
        ms_k, n_launch, bytes_step, rays_k = kernels[dom]
        launches_per_step = n_launch / args.steps
        avg_ms = ms_k / max(n_launch, 1)
        achieved = (bytes_step / launches_per_step) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": f"Mray/s on {workload_name} 1080p (path integrator, rays = Scene::Intersect + IntersectP calls)",
            "value": round(mray, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("scenes/killeroo-simple.pbrt (the reference's shipped scene file)" if args.workload == "killeroo"
                     else "synthetic scene generated by tests/boxroom.py (seed 12111)") + "; Halton samples generated on device",
            "config": {
                "workload": f"{workload_name} {args.xres}x{args.yres}, {total_spp} spp ({args.spp} per GPU), "
                            f"path maxdepth 5, halton, box filter, 16x16 tiles interleaved over {world} rank(s)",
                "xres": args.xres, "yres": args.yres, "spp_total": total_spp, "spp_per_gpu": args.spp,
                "passes_per_step": st["n_passes"],
            },
            "msamples_per_s": round(cam * args.steps / elapsed / 1e6, 3),
            "rays_per_step": rays_step,
            "camera_samples_per_step": cam,
            "rays_per_camera_sample": round(rays_step / max(cam, 1), 4),
            "n_node_per_ray": round(n_node, 3),
            "n_tri_per_ray": round(n_tri, 4),
            "b_ray_bytes": round(b_ray, 1),
            "job_algorithmic_gbs": round(mray * 1e6 * b_ray / 1e9, 1),
            "job_frac_of_hbm_roofline": round(mray * 1e6 * b_ray / 1e9 / (HBM_PEAK_GBS * world), 4),
            "kernel_ms_per_step_rank0": {k: round(agg[k] / args.steps, 3) for k in
                                         ("ms_generate", "ms_extend", "ms_shade", "ms_shadow", "ms_mis", "ms_resolve", "ms_film", "ms_total")},
            "roofline": {
                "kernel": dom,
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launches_per_step": launches_per_step,
                "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": int(bytes_step / max(launches_per_step, 1)),
                "rays_per_launch": int(rays_k / max(launches_per_step, 1)),
                "peak_measured_copy_gbs": round(copy_gbs, 1),
                "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                "kernel_choice": "the BVH traversal kernel family (extend / shadow / MIS: one traversal code, "
                                 "61 % of the step) priced on its longest member; k_shade is VALU bound, see "
                                 "roofline_all_kernels",
                "note": "algorithmic bytes = 32 B/node visited + 48 B/triangle test + 48 B/ray queue traffic "
                        "(SURVEY.md 8d), counted by the instrumented kernels; `traffic` = PMC HBM bytes per launch "
                        "(2*FETCH_SIZE + WRITE_SIZE): the BVH and mesh (~7 MB) are cache resident, so real HBM "
                        "traffic is the queue traffic and sits far below the algorithmic bytes",
            },
        }
        out["roofline_all_kernels"] = per_kernel
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.scene, args.xres, args.yres, args.cpu_seconds, workload_name)
            out["speedup_vs_cpu_baseline"] = round(mray / max(out["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
