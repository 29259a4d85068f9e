#!/usr/bin/env python3
"""bench.py — Mray/s and samples/s of the HIP path tracer on killeroo-simple 1080p.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

A *step* is one complete pass of the hot path over the configured workload:
SamplerIntegrator::Render of killeroo-simple at 1920x1080 through the C ABI (iile_render), 16x16 tiles
shared out over the N ranks by iile_tile_owner (one process per GPU; for N > 1 the driver launches this
file through torch.distributed.run), finished by ONE sum-reduction of the {X,Y,Z,w} film to rank 0 over
RCCL through the C ABI (iile_dist_film_reduce). The film stays in HBM for the whole timed region.

Workloads (BASELINE.json configs):
  N = 1                      config 2: 1920x1080 x 64 spp on one GPU (the configuration the metric is quoted on)
  N > 1 (default: strong)    config 3: the fixed 1920x1080 x 1024 spp frame, its tiles shared out over the N ranks —
                             the SAME frame at every N, so that the N = 1, 2, 4, 8 values are one scaling curve
                             (`"scaling": "strong"`; rays/s is the unit at every N, and one GPU renders 1024 spp at the
                             rate it renders 64: `other_mode` of the N = 1 line)
  --scaling weak             1920x1080 x 128*N spp, every rank renders its 1/N of the tiles at 128*N spp:
                             per-GPU work fixed; N = 8 is config 3 again
The primary mode's number is `value` over the full `--steps`; the other mode is measured too (a few steps,
`--other-steps 0` turns it off) and reported under `other_mode`.

Rank 0 prints one JSON line. `value` is whole-job Mray/s over the rays the timed kernels traced (a ray = one
Scene::Intersect or Scene::IntersectP call of the reference; MIS rays that provably cannot end on the sampled light are
not traced and not counted — `value_reference_ray_equivalents` counts every call the reference makes). `roofline` prices the kernel with the largest HIP-event time against the
HBM roof by its algorithmic bytes and by the PMC-counted HBM traffic, and says what actually binds it (VALU
issue) with the counters committed under profiles/; `cpu_baseline` is the CPU oracle timed on this box's host
cores on a bounded sample of the same workload (baseline only).

The timed region verifies itself: the film of the last timed step must equal, bit for bit, the film of the
instrumented step (different kernel builds: counting, binary BVH steps, separate camera-ray generation), and the
run refuses to start with any IILE_DEBUG_* / IILE_NO_* switch in the environment.
"""
import argparse
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0    # aggregate L2 bandwidth, same guide ("L2 (per XCD)")
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12  # CUs x SIMDs x lanes per clock x max clock: 78.6 T lane-ops/s


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


# Per-thread rate of the reference binary relative to this port, from the one measurement that exists of both on the
# same machine (the reference's own build rendered C1 at 5.8 Mray/s on 8 threads of the survey
# container; the oracle renders the same frame at 3.9 Mray/s there).
PORT_VS_REFERENCE = 3.9 / 5.8


def effective_cpus():
    """Host threads this process can really keep busy: os.cpu_count() capped by the affinity mask and by the cgroup's CPU
    quota (cpu.max = "quota period": the GPU boxes of this pool show 256 hardware threads and grant 16 CPUs' worth of time —
    profiles/r03_cpu_scaling.json: the oracle scales 7.9x on 8 threads, 13.6x on 16 and is throttled beyond)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def cpu_baseline(scene_path, xres, yres, target_seconds, workload_name="killeroo-simple"):
    """Time the CPU oracle (restatement of the reference path, 16x16 tile self-scheduling, render loop only) on a bounded
    number of pixel samples, on as many threads as the box really grants (effective_cpus)."""
    import __graft_entry__ as ge
    import oracle_binding as ob
    b = ge._load_binding()
    threads, quota = effective_cpus()
    scene = b.HostScene(path=scene_path, xres=xres, yres=yres, spp=64)
    orc = ob.Oracle()
    _, st = orc.render(scene, trig_mode=ob.TRIG_LIBM, threads=threads, k_begin=0, k_end=1)
    t1 = max(st["seconds"], 1e-3)
    n = int(max(1, min(63, round(target_seconds / t1))))  # (the scene has 64 samples per pixel; sample 0 was the probe)
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
        "host": {"hardware_threads": os.cpu_count(), "cgroup_cpu_quota": quota},
        "note": f"a port, not the reference binary (which cannot be built in this image): on the one machine where "
                f"both were timed (BASELINE.md §2, 8 threads) the port ran {PORT_VS_REFERENCE:.2f}x the reference's rate, "
                f"so the reference on these cores would be about {1 / PORT_VS_REFERENCE:.2f}x this value. Threads = what the "
                f"box grants: the pool's hosts show {os.cpu_count()} hardware threads but a cgroup quota of "
                f"{quota if quota else 'none'} CPUs; the measured thread curve (profiles/r03_cpu_scaling.json, tools/cpu_scaling.py) is "
                f"linear up to that (7.9x on 8 threads, 13.6x on 16) and throttled beyond — rounds 1 and 2 ran 256 threads "
                f"against that quota and quoted the throttled figure as '256 cores'",
        "value_scaled_to_reference": round(rays / st["seconds"] / 1e6 / PORT_VS_REFERENCE, 3),
    }


def load_pmc(world, args, total_spp):
    """The committed counter summaries (tools/summarize_profiles.py): newest profiles/*_pmc_traffic.json and
    *_pmc_lanes.json of this workload on one GPU (killeroo-simple: the default frame; boxroom: files tagged `_room`)."""
    # (the counters were collected on the 64-spp step: they say nothing about a step of another size, e.g. --scaling strong)
    if not (world == 1 and (args.xres, args.yres, total_spp) == (1920, 1080, 64) and args.workload in ("killeroo", "boxroom") and not args.sampler):
        return {}, {}, None, None
    room = args.workload == "boxroom"

    def newest(pattern):
        for f in sorted(glob.glob(os.path.join(REPO, "profiles", pattern)))[::-1]:
            if ("_room" in os.path.basename(f)) != room:
                continue
            try:
                j = json.load(open(f))
                if "families" in j:
                    return j["families"], os.path.basename(f)
            except Exception:
                pass
        return {}, None

    tr, tr_src = newest("*_pmc_traffic.json")
    la, la_src = newest("*_pmc_lanes.json")
    return tr, la, tr_src, la_src


def load_vmem_calibration():
    """The measured ceiling of the vector-memory path for the traversal's access shape (tools/vmem_calib.hip: dependent
    fetches of 128-byte records, 7 loads of 16 bytes per lane and step, 24 wavefronts per CU): lane-loads per ns per CU."""
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_vmem_calib.json")))[::-1]:
        try:
            j = json.load(open(f))
            rows = [r for r in j["rows"] if r["mode"] == 0 and r["pieces"] == 7 and r["active"] == 64 and r["blocks_per_cu"] == 6]
            by_mb = {int(r["table_mb"]): r["lane_loads_per_ns_cu"] for r in rows}
            if by_mb:
                return by_mb, os.path.basename(f)
        except Exception:
            pass
    return {}, None


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE config 5: the IISPT integrator's frame (--workload iispt)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 / f16 matrix peak (the F16 forms take the same cycles), /opt/skills/guides/MI355X_MICROARCH.md ("~2.5 PF dense")
MFMA_F32_PEAK_TFLOPS = 157.3     # fp32-input matrix instructions run at the vector rate (same guide)
# IISPTNet.forward per probe (ml/iispt_net.py:8-109): 2 * k * k * C_in * C_out * H * W over its 15 convolutions
NET_FLOP_PER_PROBE = 2 * 9 * (7 * 64 * 1024 + 64 * 64 * 1024 + 64 * 128 * 256 + 128 * 128 * 256 + 128 * 256 * 64 + 256 * 256 * 64
                              + 256 * 512 * 16 + 512 * 256 * 16 + 512 * 256 * 64 + 256 * 128 * 64 + 256 * 128 * 256 + 128 * 64 * 256
                              + 128 * 64 * 1024 + 64 * 64 * 1024) + 2 * 64 * 3 * 1024


def iispt_cpu_baseline(b, ref_mod, frame_mod, scene, net_module, target_seconds):
    """The reference's per-probe loop on this box's host, on a bounded sample: the CPU oracle's hemi points of the frame's
    first task(s), one oracle probe render per point, the network ONE probe at a time on ONE thread (as
    ml/main_stdio_net.py:104-118 runs it: torch.set_num_threads(1), one `net(x)` per request), the oracle's gather."""
    import numpy as np
    import torch
    import oracle_binding as ob
    orc = ob.Oracle()
    h, w = scene.film_shape
    torch.set_num_threads(1)
    net_module = net_module.cpu().eval()
    t_probe = t_net = t_gather = t_hemi = 0.0
    n_probes = n_pixels = n_tasks = 0
    counter = 0
    t_start = time.perf_counter()
    for (x0, y0, x1, y1, ts) in frame_mod.schedule((0, 0, w, h), 10 ** 6, 10.0):
        task = b.IisptTask(x0, y0, x1, y1, ts, counter, 0)
        nx, ny = task.grid()
        t0 = time.perf_counter()
        valid, pos, dr = orc.iispt_hemi_points(scene, task, trig_mode=ob.TRIG_LIBM)
        t_hemi += time.perf_counter() - t0
        nn_films = np.zeros((ny * nx, 32, 32, 3), np.float32)
        for i in np.flatnonzero(valid.reshape(-1) == 1):
            t0 = time.perf_counter()
            inten, nrm, dist = orc.render_probe(scene, pos.reshape(-1, 3)[i], dr.reshape(-1, 3)[i], trig_mode=ob.TRIG_LIBM)
            t1 = time.perf_counter()
            with torch.no_grad():
                x, means = ref_mod.normalize_downstream(torch.from_numpy(inten[None]), torch.from_numpy(nrm[None]), torch.from_numpy(dist[None]))
                pred = ref_mod.transform_upstream(net_module(x), means)
            nn_films[i] = pred[0].numpy()[::-1]
            t2 = time.perf_counter()
            t_probe += t1 - t0
            t_net += t2 - t1
            n_probes += 1
        t0 = time.perf_counter()
        orc.iispt_gather(scene, task, valid, pos, dr, nn_films, trig_mode=ob.TRIG_LIBM)
        t_gather += time.perf_counter() - t0
        n_pixels += (x1 - x0) * (y1 - y0)
        n_tasks += 1
        counter += nx * ny + (x1 - x0) * (y1 - y0)
        if time.perf_counter() - t_start > target_seconds:
            break
    total = t_hemi + t_probe + t_net + t_gather
    return {"value": round(n_probes / max(total, 1e-9), 3), "unit": "probes/s", "cores": 1, "kind": "port",
            "sample": f"the first {n_tasks} task(s) of the frame's schedule (100 x 100 pixels each): {n_probes} probes, {n_pixels} pixels gathered, "
                      f"{total:.1f} s on one thread (hemi points {t_hemi:.2f} s, probe renders {t_probe:.2f} s, network + transforms {t_net:.2f} s, gather {t_gather:.2f} s)",
            "ms_per_probe": {"probe_render": round(t_probe / max(n_probes, 1) * 1e3, 2), "network_and_transforms": round(t_net / max(n_probes, 1) * 1e3, 2)},
            "reference_quoted_ms_per_probe": {"network_round_trip": 47, "network_alone": 27, "source": "Doc.md:55-64 (the reference authors' machine)"},
            "note": "a port, not the reference binary: the CPU oracle (oracle/) for hemi points, probe renders and the gather, and the PyTorch module on "
                    "ONE thread, one probe per call, for the network — the reference's IISPT runner is one process per render thread piping each "
                    "probe to a single-threaded Python child (ml/main_stdio_net.py:106); its other threads would scale this by the host's cores"}


def main_iispt(args):
    """One step = one IISPT frame over killeroo-simple at args.xres x args.yres (default 1080p): the indirect pass at radius 10
    (IisptRenderRunner::run over one sweep of the schedule: hemi points, probe pass, network, gather), the direct pass (16 passes
    of DirectProgressiveIntegrator) and the merge of the two film monitors (pbrt-v3-iile_amd/iispt_frame.py). Prints the
    contract's line: value = probes per second over the whole frame; roofline = the network's convolution kernels against the
    16-bit matrix peak; cpu_baseline = the reference's per-probe loop on one host thread on a bounded sample."""
    import importlib
    import numpy as np
    import torch
    import __graft_entry__ as ge
    compiled_now = ge.build_if_needed()
    b = ge._load_binding()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path has no CPU fallback")
    # N > 1 (strong scaling of the ONE frame): tasks dealt by their number, direct passes in contiguous blocks, one all-reduce (RCCL)
    # per film monitor — iispt_frame.py; the reference's render threads draw tasks and pass numbers from one schedule monitor
    # (iispt.cpp:386-427), and a task / a pass is the same whoever renders it
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --workload iispt --gpus {args.gpus} ...`")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    nn_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_nn")
    import iispt_torch_reference as ref_mod   # tests/: the PyTorch module (random weights for the frame; the checker of the in-run agreement test)
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    scene = b.HostScene(path=args.scene, xres=args.xres, yres=args.yres, spp=1)
    gpu = b.GpuScene(scene)
    torch.manual_seed(0)
    module = ref_mod.IISPTNet().eval()   # no trained weights ship with the reference: random-initialised, same architecture and cost
    pipe = nn_mod.IisptPipeline(gpu, net=module, binding=b, batch=args.net_batch)
    radius = 10.0
    size = int(radius) * frame_mod.NUMBER_TILES
    n_tasks = -(-args.xres // size) * -(-args.yres // size)

    def step(record=None):
        pipe.events = record
        frame = frame_mod.IisptFrame(b, gpu, pipe)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        frame.run_batched(n_tasks, radius_start=radius, rank=rank, nranks=world)
        ev[1].record()
        frame.run_direct(frame_mod.DIRECT_SAMPLES, rank=rank, nranks=world)
        ev[2].record()
        frame.reduce_monitors(dist)   # (N = 1: nothing to add)
        img = frame.image()
        ev[3].record()
        pipe.events = None
        return frame, img, ev

    for _ in range(max(args.warmup, 1)):   # (the first frame allocates the workspaces)
        step()
    barrier()
    t0 = time.perf_counter()
    last = None
    stage_events, frame_events = [], []
    for _ in range(args.steps):
        rec = []
        last = step(rec)
        stage_events.append(rec)
        frame_events.append(last[2])
    barrier()
    elapsed = time.perf_counter() - t0
    frame, img, _ = last
    probes = frame.stats["probes"]
    if dist is not None:   # the slowest rank's time; every rank's probes (a rank's statistics cover its own tasks)
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n = torch.tensor([probes, frame.stats["hemi_points"], frame.stats["pixels"]], dtype=torch.int64, device="cuda")
        per_rank = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(per_rank, n)
        n = sum(per_rank)
        probes, frame.stats["hemi_points"], frame.stats["pixels"] = int(n[0]), int(n[1]), int(n[2])
        rank_probes = [int(x[0]) for x in per_rank]
        if rank != 0:
            dist.barrier()
            dist.destroy_process_group()
            return None
    stage_ms = {}
    for rec in stage_events:
        for name, e0, e1 in rec:
            stage_ms[name] = stage_ms.get(name, 0.0) + e0.elapsed_time(e1) / args.steps
    indirect_ms = sum(e[0].elapsed_time(e[1]) for e in frame_events) / args.steps
    for name in ("normalize", "rescale"):   # (inside `network` with the HIP backend: iile_iispt_net_predict runs the two transforms)
        stage_ms.setdefault(name, 0.0)
    stage_ms["hemi_points_gather_and_film"] = indirect_ms - sum(stage_ms.values())
    stage_ms["direct_pass_16"] = sum(e[1].elapsed_time(e[2]) for e in frame_events) / args.steps
    stage_ms["merge"] = sum(e[2].elapsed_time(e[3]) for e in frame_events) / args.steps
    # the network alone, over the frame's probes: HIP events around iile_iispt_net_forward on its stream (the `network` stage)
    net_ms = stage_ms["network"]
    net_launches = sum(1 for n_, _, _ in stage_events[-1] if n_ == "network")
    flop = NET_FLOP_PER_PROBE * probes
    ach = flop / (net_ms * 1e-3) / 1e12
    # memory-side bytes of a forward from the committed counter passes (tools/net_traffic.sh: FETCH_SIZE doubled + WRITE_SIZE of every
    # network kernel at 8 192 probes, per probe) x this frame's probes
    net_traffic, net_traffic_note = None, "no committed counter set (profiles/*_net_traffic.json)"
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "*_net_traffic.json")))[::-1]:
        try:
            tj = json.load(open(f))
            net_traffic = int(tj["per_probe_bytes"] * probes)
            net_traffic_note = (f"profiles/{os.path.basename(f)}: {tj['per_probe_bytes'] / 1e6:.2f} MB per probe (activations written once: 2.44 MB; read once: "
                                f"2.88 MB; the rest is halo rows, tiles re-read per 64 output channels, weights) = "
                                f"{net_traffic / (net_ms * 1e-3) / 1e9 / HBM_PEAK_GBS:.3f} of the 8 TB/s HBM peak at this run's network time")
            break
        except Exception:
            pass
    # agreement of what was timed — iile_iispt_net_predict: normalizeMapsDownstream, the network, transformMapsUpstream — with the PyTorch
    # statement on the CPU (tests/iispt_torch_reference.py), on probes of the timed frame itself: the first valid hemi points of its first task
    with torch.no_grad():
        x0, y0, x1, y1, ts = next(iter(frame_mod.schedule((0, 0, args.xres, args.yres), 1, radius)))
        task0 = b.IisptTask(x0, y0, x1, y1, ts, 0, 0)
        valid, hp, hd = gpu.iispt_hemi_points_batch([task0])
        sel = np.flatnonzero(valid == 1)[:16]
        pred, inten, nrm, dst = pipe(hp[sel], hd[sel])
        xr, means = ref_mod.normalize_downstream(inten.cpu(), nrm.cpu(), dst.cpu())
        want = ref_mod.transform_upstream(module(xr), means).double().numpy().ravel()
        got = pred.cpu().double().numpy().ravel()
        mx = float(np.abs(want).max())
        err = np.abs(got - want)
        chk = {"probes": int(len(sel)), "what": "iile_iispt_net_predict vs normalize_downstream -> IISPTNet (fp32, CPU) -> transform_upstream on the first valid "
                                                 "hemi points of the timed frame's first task",
               "max_abs_err_over_max": float(err.max() / max(mx, 1e-30)), "bound": 1e-5,
               "elements_within_1e-4_rel_plus_1e-6_of_max": float((err <= 1e-4 * np.abs(want) + 1e-6 * mx).mean()),
               "mean_rel_err_where_nonzero": float((err[np.abs(want) > 1e-6 * mx] / np.abs(want[np.abs(want) > 1e-6 * mx])).mean())}
    if not (chk["max_abs_err_over_max"] < 1e-5 and chk["elements_within_1e-4_rel_plus_1e-6_of_max"] >= 0.999) or not bool(torch.isfinite(img).all()):
        raise SystemExit(f"bench.py: the timed network disagrees with the PyTorch module ({chk}) or the frame is not finite; no number is reported")
    out = {
        "metric": "IISPT probes/s on killeroo-simple 1080p (one frame: hemi points + probe pass + network + gather, direct pass, merge)",
        "value": round(probes * args.steps / elapsed, 1),
        "unit": "probes/s",
        "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
        "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
        "dtype": "f32 activations and accumulation; matrix products on fp16 pairs (hi + lo: 22 significant bits) of every operand",
        "data": "scenes/killeroo-simple.pbrt; IISPTNet with random-initialised weights (none ship with the reference): the image means nothing, the work is the reference's",
        "config": {"workload": f"IISPT frame, killeroo-simple {args.xres}x{args.yres}: radius 10 -> {n_tasks} tasks of 100 x 100 pixels, {frame.stats['hemi_points']} hemi points, "
                               f"{probes} probes of 32 x 32, every pixel gathered from 4 probes; 16 direct passes; merge",
                   "baseline_config": "5 (IISPT integrator: hemisphere probes + network on the GPU)", "xres": args.xres, "yres": args.yres,
                   "probes": probes, "pixels": frame.stats["pixels"]},
        "stage_ms_per_step": {k: round(v, 3) for k, v in stage_ms.items()},
        "stage_note": "HIP events on the stream every stage runs on; `network` = iile_iispt_net_predict: normalizeMapsDownstream, the 15 convolutions, "
                      "transformMapsUpstream (`normalize` / `rescale` are 0: they are kernels of that call); `hemi_points_gather_and_film` = the "
                      "indirect pass minus its timed stages",
        "roofline": {
            "kernel": "k_conv3x3 (the 14 3x3 convolutions of IISPTNet, csrc/device/iispt_net.hip; with the two layout kernels of a forward)",
            "bound": "mfma",
            "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
            "traffic": net_traffic,
            "traffic_note": net_traffic_note,
            "algorithmic_flop_per_unit": NET_FLOP_PER_PROBE, "units_per_step": probes,
            "algorithmic_note": "2 k^2 C_in C_out H W over the network's 15 convolutions = 0.990 GFLOP per probe (fp32 multiply-adds of the reference's "
                                "module); `achieved` = that x probes / the network's HIP-event time",
            "executed_f16_tflops": round(3 * ach, 1), "frac_executed": round(3 * ach / MFMA_BF16_PEAK_TFLOPS, 4),
            "executed_note": "every product runs as three f16 matrix instructions (a_hi w_hi + a_hi w_lo + a_lo w_hi; same cycles as the bf16 forms), so the matrix pipe executes 3x the "
                             "algorithmic flops (the first layer also multiplies 9 zero-padded input channels)",
            "vs_fp32_matrix_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 3),
            "vs_fp32_matrix_peak_note": "the fp32-input matrix instructions peak at 157.3 TFLOP/s on gfx950: a ratio above 1 is what the split buys",
            "network_ms_per_step": round(net_ms, 3), "forward_calls_per_step": net_launches,
            "agreement_with_the_module": chk},
        "built": ge.build_provenance(compiled_now),
    }
    if world > 1:
        out["per_rank"] = {"probes": rank_probes, "note": "tasks dealt by their number (rank = task mod N), the 16 direct passes in contiguous blocks, one RCCL all-reduce per "
                                                             "film monitor (2 x 66 MB of doubles at 1080p); stage times above are rank 0's share"}
        dist.barrier()
        dist.destroy_process_group()
    if args.cpu_seconds > 0 and world == 1:
        out["cpu_baseline"] = iispt_cpu_baseline(b, ref_mod, frame_mod, scene, module, args.cpu_seconds)
        out["speedup_vs_one_cpu_thread"] = round(out["value"] / max(out["cpu_baseline"]["value"], 1e-9), 1)
        threads, _q = effective_cpus()
        out["speedup_vs_cpu_baseline_scaled_to_granted_cpus"] = round(out["value"] / max(out["cpu_baseline"]["value"] * threads, 1e-9), 1)
        out["speedup_note"] = (f"the baseline is ONE thread of the per-probe loop; the reference runs one runner + one Python child per core, so the second "
                               f"figure divides by the baseline x the {threads} CPUs this box grants (linear scaling assumed: generous to the CPU)")
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=0,
                    help="weak mode: pixel samples per GPU-equivalent (total = spp * gpus); default 64 at 1 GPU "
                         "(BASELINE config 2), 128 at N > 1 (N = 8: config 3's 1024 spp)")
    ap.add_argument("--strong-spp", type=int, default=1024, help="strong mode: pixel samples of the fixed frame (config 3)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: weak at N = 1 (config 2), strong at N > 1 (config 3's fixed 1024 spp frame)")
    ap.add_argument("--other-steps", type=int, default=2, help="timed steps of the other scaling mode (0: skip it)")
    ap.add_argument("--spp-per-pass", type=int, default=0)
    ap.add_argument("--alone-steps", type=int, default=2,
                    help="untimed extra steps with every kernel on one stream, for per-kernel durations without overlap (0: skip)")
    ap.add_argument("--schedule", choices=["two-stream", "one-stream"], default="two-stream",
                    help="two-stream: the product's schedule (the NEE kernels of a bounce beside the next bounce's extend / shade); "
                         "one-stream: every kernel of the timed steps alone on the GPU — for kernel traces whose per-kernel averages are "
                         "not mixed with time spent sharing the chip (profiles/*_one_stream_kernel_stats.csv); `value` is then the "
                         "one-stream schedule's and says so")
    ap.add_argument("--scene", default=os.path.join(REPO, "scenes", "killeroo-simple.pbrt"))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the cpu_baseline leg")
    ap.add_argument("--sampler", choices=["halton", "sobol"], default=None,
                    help="sampler in place of the scene file's (sobol: what the fork's path integrator renders with under "
                         "IILE_PATH_SAMPLES_OVERRIDE; the headline metric is quoted with the scene's own Halton sampler)")
    ap.add_argument("--workload", choices=["killeroo", "boxroom", "boxroom-textured", "iispt"], default="killeroo",
                    help="boxroom: the synthetic ~287 k-triangle closed room of tests/boxroom.py (deep-BVH stress, "
                         "SURVEY.md 8d's stand-in for the Sponza config that does not ship with the reference); "
                         "boxroom-textured: the same room open to an environment-mapped sky, with image textures, "
                         "alpha masks and specular materials (the whole feature set of SURVEY.md 8 f1); "
                         "iispt: BASELINE config 5, one frame of the IISPT integrator (probe pass, network, gather, direct pass)")
    ap.add_argument("--net-batch", type=int, default=8192, help="--workload iispt: probes per set of network launches (iile_iispt_net_predict's max_batch)")
    ap.add_argument("--sub-configs", default="4_room,5_iispt",
                    help="with the default workload on one GPU: BASELINE configs measured after the headline steps and printed as sub-blocks "
                         "`configs` of the same JSON line (4_room: the deep-tree room, 1080p x 64 spp; 5_iispt: one IISPT frame at 1080p); "
                         "`none` turns them off")
    ap.add_argument("--sub-steps", type=int, default=0, help="timed steps of each sub-config (default: 2 for the room, 3 for the IISPT frame)")
    ap.add_argument("--sub-cpu-seconds", type=float, default=6.0, help="cpu_baseline budget of each sub-config")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    bad_env = sorted(k for k in os.environ if k.startswith("IILE_DEBUG") or k.startswith("IILE_NO_"))
    if bad_env:
        raise SystemExit(f"bench.py refuses to run with {bad_env} set: those switches change what the kernels do")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.workload == "iispt":
        line = main_iispt(args)
        if line is not None:   # (rank 0 prints)
            print(json.dumps(line), flush=True)
        return
    out = main_path(args)
    subs = [x for x in args.sub_configs.split(",") if x and x != "none"]
    if out is not None and subs and world == 1 and args.workload == "killeroo" and (args.xres, args.yres) == (1920, 1080) and not args.sampler:
        out["configs"] = sub_configs(args, subs)
    if out is not None:
        print(json.dumps(out), flush=True)


def sub_configs(args, subs):
    """BASELINE configs 4 and 5 in the line the driver runs (VERDICT r05 "next" 1): after the headline's steps, the deep-tree room
    (2 steps) and one IISPT frame (3 steps), each a whole line of its own workload — ms_per_step, roofline, cpu_baseline, the in-run
    parity check — nested under `configs`. The headline keys are config 2's and do not change."""
    import torch
    blocks = {}
    for name in subs:
        a = argparse.Namespace(**vars(args))
        a.cpu_seconds = args.sub_cpu_seconds if args.cpu_seconds > 0 else 0.0
        a.other_steps = 0
        t0 = time.perf_counter()
        if name == "4_room":
            a.workload, a.steps, a.warmup = "boxroom", args.sub_steps or 2, 1
            blk = main_path(a)
        elif name == "5_iispt":
            a.workload, a.steps, a.warmup = "iispt", args.sub_steps or 3, 1
            blk = main_iispt(a)
        else:
            raise SystemExit(f"--sub-configs: unknown block {name!r} (4_room, 5_iispt, none)")
        blk.pop("built", None)   # (the line's own `built` covers the one set of libraries this process runs)
        blk["wall_seconds_of_this_block"] = round(time.perf_counter() - t0, 1)
        blocks[name] = blk
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return blocks


def main_path(args):
    """The path integrator's line (configs 2 / 3 / 4): returns the dict rank 0 prints (None on the other ranks)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")
    if args.spp <= 0:
        args.spp = 64 if world == 1 else 128
    if args.scaling is None:
        args.scaling = "weak" if world == 1 else "strong"

    cleanup = []
    workload_name = "killeroo-simple"
    if args.workload in ("boxroom", "boxroom-textured"):
        import atexit
        import shutil
        import tempfile
        import boxroom
        tmp = tempfile.NamedTemporaryFile("w", suffix=".pbrt", delete=False)
        cleanup.append(tmp.name)
        atexit.register(lambda: [shutil.rmtree(p, ignore_errors=True) if os.path.isdir(p) else (os.path.exists(p) and os.remove(p))
                                 for p in cleanup])
        if args.workload == "boxroom":
            tmp.write(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64))
            workload_name = "synthetic boxroom (287k triangles, tests/boxroom.py)"
        else:
            texdir = tempfile.mkdtemp(prefix="boxroom_img_")
            cleanup.append(texdir)
            tmp.write(boxroom.boxroom_pbrt(ico_levels=5, n_blobs=12, wall_n=64, light="envmap", materials="mixed", textures=texdir))
            workload_name = "synthetic textured boxroom (266k triangles, environment map, image textures, alpha masks; tests/boxroom.py)"
        tmp.close()
        args.scene = tmp.name

    import numpy as np
    import torch
    import __graft_entry__ as ge
    compiled_now = ge.build_if_needed()
    b = ge._load_binding()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist, comm = None, None
    import importlib.util
    spec = importlib.util.spec_from_file_location("iile_multigpu", os.path.join(REPO, "pbrt-v3-iile_amd", "multigpu.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        comm = mg.create_comm(dist, f"cuda:{local_rank}")  # the film merge goes through the C ABI (libiile_dist.so)
    # libiile_dist.so is linked against librccl by SONAME and torch ships its own copy: two RCCL instances in one
    # process would each think they own the GPUs' IPC state. Exactly one may be mapped.
    rccl_mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
    if world > 1 and len({os.path.realpath(x) for x in rccl_mapped}) != 1:
        raise SystemExit(f"bench.py: libiile_dist and torch resolved different RCCL libraries: {rccl_mapped}")
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(total_spp, steps, warmup, want_kernels):
        """Instrumented step, warm-up, `steps` timed steps between barriers; returns the job's numbers (rank 0)."""
        scene = b.HostScene(path=args.scene, xres=args.xres, yres=args.yres, spp=total_spp, sampler=args.sampler)
        gpu = b.GpuScene(scene)
        h, w = scene.film_shape
        film = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")

        rank_times = []  # per timed step: (render ms, film-reduce ms) of THIS rank, from events on the render stream

        def step(collect=False, timed=False, clock=False):
            box = {}
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if clock else None

            def render(tile_rank, tile_nranks):
                if ev:
                    ev[0].record()
                _, box["st"] = gpu.render(tile_rank=tile_rank, tile_nranks=tile_nranks, spp_per_pass=args.spp_per_pass,
                                          collect_stats=collect, time_kernels=timed, film_device_ptr=film.data_ptr(),
                                          stream=stream, want_stats=True)
                if ev:
                    ev[1].record()

            mg.render_sharded(render, film, dist, comm=comm, stream=stream)  # N > 1: one RCCL sum-reduction to rank 0
            if ev:
                ev[2].record()
                rank_times.append(ev)
            return box["st"]

        # instrumented step (untimed): ray / node / triangle counts of one step on this rank, and the film every
        # timed step must reproduce bit for bit
        cst = step(collect=True)
        torch.cuda.synchronize()
        film_check = film.clone()
        for _ in range(warmup):
            step(timed=2 if (args.schedule == "one-stream" and want_kernels) else False)   # (a trace of the run then holds one schedule only)
        barrier()
        t0 = time.perf_counter()
        agg = {"ms_extend": 0.0, "ms_connect": 0.0, "ms_shadow": 0.0, "ms_mis": 0.0, "ms_resolve": 0.0, "ms_shade": 0.0,
               "ms_generate": 0.0, "ms_film": 0.0, "ms_total": 0.0, "n_extend_launches": 0, "n_connect_launches": 0,
               "n_shade_launches": 0}
        st = None
        for _ in range(steps):
            st = step(timed=(2 if args.schedule == "one-stream" else 1) if want_kernels else False, clock=True)
            for k in agg:
                agg[k] += st[k]
        barrier()
        elapsed = time.perf_counter() - t0
        # this rank's share of a step: its render (iile_render on its tiles) and its part of the one film reduction
        ms_render = sum(e[0].elapsed_time(e[1]) for e in rank_times) / max(len(rank_times), 1)
        ms_reduce = sum(e[1].elapsed_time(e[2]) for e in rank_times) / max(len(rank_times), 1)
        # untimed extra steps with every kernel alone on the GPU (one stream): the per-kernel durations of the timed
        # steps overlap (the NEE kernels of a bounce run beside the next bounce's extend / shade on a second stream)
        alone = {k: 0.0 for k in agg}
        n_alone = args.alone_steps if want_kernels else 0
        for _ in range(n_alone):
            st1 = step(timed=2)
            for k in alone:
                alone[k] += st1[k]
        torch.cuda.synchronize()
        # self-verification of the timed region: every rank's last timed film (rank 0: the merged film) against the
        # instrumented step's
        same = bool(torch.equal(film.view(torch.int32), film_check.view(torch.int32)))
        # rays the timed kernels traced on this rank: every main-path and shadow ray, and the MIS rays the plain build
        # does not prove irrelevant (counted by the timed step itself)
        traced = st["ext_rays_traced"] + cst["shadow_rays"] + st["mis_rays_traced"]
        per_rank = {"ms_render": [ms_render], "ms_reduce": [ms_reduce], "rays_traced": [int(traced)]}
        if dist is not None:
            mine = torch.tensor([ms_render, ms_reduce, float(traced)], dtype=torch.float64, device="cuda")
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank = {"ms_render": [float(x[0]) for x in every], "ms_reduce": [float(x[1]) for x in every],
                        "rays_traced": [int(x[2]) for x in every]}
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            cnt = torch.tensor([cst["closest_rays"], cst["shadow_rays"], cst["camera_rays"], traced, 1 if same else 0], dtype=torch.int64, device="cuda")
            mn = cnt[4:].clone()
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            rays_closest, rays_shadow, cam, traced = (int(x) for x in cnt[:4].tolist())
            same = bool(int(mn.item()))
        else:
            rays_closest, rays_shadow, cam = cst["closest_rays"], cst["shadow_rays"], cst["camera_rays"]
        if not same:
            raise SystemExit("bench.py: the film of the last timed step differs from the instrumented step's film — "
                             "the timed kernels did not do the reference's work; no number is reported")
        rays = rays_closest + rays_shadow
        res = {"elapsed": elapsed, "steps": steps, "rays_step": rays, "rays_traced_step": traced, "cam": cam, "cst": cst, "agg": agg, "n_passes": st["n_passes"],
               "per_rank": per_rank,
               "alone": alone, "n_alone": n_alone,
               "ms_per_step": elapsed * 1e3 / steps, "mray": traced * steps / elapsed / 1e6,
               "mray_reference": rays * steps / elapsed / 1e6, "total_spp": total_spp}
        del gpu, scene, film, film_check
        torch.cuda.empty_cache()
        return res

    modes = {"weak": args.spp * world, "strong": args.strong_spp}
    primary = measure(modes[args.scaling], args.steps, args.warmup, True)
    other_name = "strong" if args.scaling == "weak" else "weak"
    other = None
    if args.other_steps > 0 and args.workload == "killeroo" and modes[other_name] != modes[args.scaling]:
        other = measure(modes[other_name], args.other_steps, 1, False)

    if rank == 0:
        cst, agg = primary["cst"], primary["agg"]
        steps = args.steps
        rays_step, cam, mray = primary["rays_step"], primary["cam"], primary["mray"]
        total_spp = primary["total_spp"]
        # per-ray averages of this rank's instrumented step
        r_all = cst["closest_rays"] + cst["shadow_rays"]
        n_node = (cst["nodes_closest"] + cst["nodes_any"]) / max(r_all, 1)
        n_tri = cst["tri_tests"] / max(r_all, 1)
        b_ray = 32 * n_node + 48 * n_tri + 48
        # algorithmic bytes per step of every pipeline kernel (this rank): SURVEY.md 8d's figure for the traversal
        # kernels; queue / path-state records for the others (the bytes that have to cross HBM: the ~7 MB scene is
        # cache resident and is not counted for them)
        mis_rays = cst["closest_rays"] - cst["ext_rays"]
        nee = cst["nee_evals"]
        next_rays = cst["ext_rays"] - cst["camera_rays"]
        fam = {
            "k_extend": (agg["ms_extend"], agg["n_extend_launches"], algorithmic_bytes(cst["ext_rays"], cst["ext_nodes"], cst["ext_tri_tests"]),
                         cst["ext_rays"], "32 B/node + 48 B/triangle test + 48 B/ray (SURVEY.md 8d)"),
            "k_shade": (agg["ms_shade"], agg["n_shade_launches"], 68 * nee + 48 * nee + 48 * next_rays, nee,
                        "per hit 68 B in (queue entry 4, hit 16, ray 32, throughput + Halton index 16), 48 B NEE record out, 48 B next ray with "
                        "its path state out per continued path (queue records only; round 2 moved 72 + 112 + 48)"),
            "k_shadow": (agg["ms_shadow"], agg["n_connect_launches"], algorithmic_bytes(cst["shadow_rays"], cst["nodes_any"], cst["any_tri_tests"]),
                         cst["shadow_rays"], "as k_extend (+ 48 B record in, 32 B L read-modify-write, not counted in SURVEY's figure)"),
            "k_mis": (agg["ms_mis"], agg["n_connect_launches"],
                      algorithmic_bytes(mis_rays, cst["nodes_closest"] - cst["ext_nodes"], cst["tri_tests"] - cst["ext_tri_tests"] - cst["any_tri_tests"]),
                      mis_rays, "as k_extend"),
            "k_mis_lit": (agg["ms_resolve"], agg["n_connect_launches"], 1 * nee, nee, "one result byte per NEE record"),
            "k_film": (agg["ms_film"], primary["n_passes"] * steps, 16 * cst["camera_rays"], cst["camera_rays"], "L read per sample"),
        }
        pmc_traffic, pmc_lanes, tr_src, la_src = load_pmc(world, args, total_spp)
        vmem_peak, vmem_src = load_vmem_calibration()
        # What the counters say binds each kernel family (profiles/*_pmc_mem.json, *_pmc_lanes.json, r04_vmem_calib.json):
        BINDS = {
            "k_extend": "vmem", "k_shadow": "vmem", "k_mis": "vmem",
            "k_shade": "registers+vmem", "k_mis_lit": "hbm", "k_film": "hbm",
        }
        BIND_NOTES = {
            "vmem": "the vector-memory path in front of the L1: the kernel retires 16-byte lane-loads at the rate tools/vmem_calib.hip "
                    "measures as the ceiling for dependent fetches of 128-byte records (address unit busy 0.6-0.86, data return 0.9-0.99 of "
                    "the kernel's cycles: profiles/*_pmc_mem.json); not HBM (the scene is cache resident), not VALU issue (0.4-0.45 of its ceiling)",
            "registers+vmem": "waves per SIMD (128 VGPRs: four) and vector-memory instructions per hit; VALU issue at ~0.6 of the calibrated ceiling (DESIGN.md section 6)",
            "hbm": "streams its records once: HBM bandwidth",
        }
        # Per-kernel durations: a kernel ALONE on the GPU — the one-stream steps after the timed region (or the timed steps themselves
        # under --schedule one-stream). In the two-stream timed steps the NEE kernels of a bounce share the chip with the next
        # bounce's extend / shade: those HIP-event durations overlap, sum to more than a step and say how long a kernel was
        # resident, not how fast it is; they are kept under `overlapped` and never summed.
        fam_key = {"k_extend": "ms_extend", "k_shade": "ms_shade", "k_shadow": "ms_shadow", "k_mis": "ms_mis", "k_mis_lit": "ms_resolve",
                   "k_film": "ms_film"}
        one_stream_timed = args.schedule == "one-stream"
        have_alone = primary["n_alone"] > 0 or one_stream_timed

        def alone_ms(k):   # ms per step of kernel family k with the GPU to itself
            if one_stream_timed:
                return agg[fam_key[k]] / steps
            return primary["alone"][fam_key[k]] / primary["n_alone"]

        per_kernel = {}
        step_counter_bytes, step_counter_kernels = 0, []
        for k, (ms_sum, n_l, by, units, note) in fam.items():
            if ms_sum <= 0:
                continue
            ms_step = alone_ms(k) if have_alone else ms_sum / steps   # ms per step this entry is priced with
            if ms_step <= 0:
                continue
            by_step = by   # (the counts come from ONE instrumented step)
            e = {"ms_per_step": round(ms_step, 3), "schedule": "one-stream (the kernel alone on the GPU)" if have_alone else "two-stream (overlapped: no one-stream steps in this run)",
                 "launches_per_step": round(n_l / steps, 2), "bound": BINDS.get(k, "hbm"),
                 "algorithmic_gbs": round(by_step / ms_step / 1e6, 1), "algorithmic_bytes_note": note}
            if have_alone and not one_stream_timed:
                e["overlapped"] = {"ms_per_step": round(ms_sum / steps, 3),
                                   "note": "HIP-event duration in the two-stream timed steps: includes time spent sharing the GPU with the other stream's "
                                           "kernels; not a kernel speed, never summed"}
            tr = pmc_traffic.get(k if k != "k_film" else "k_film_accumulate")
            if tr:
                # The counters sit at the L2's memory side: they count what the Infinity Cache serves as well as what HBM
                # serves (MI355X_MICROARCH.md, HBM section). A kernel must stream its queue records from HBM once; whatever it
                # reads beyond that is scene data (7.6 MB of four-wide records + 3.2 MB of triangles for killeroo-simple,
                # against 4 MiB of L2 per XCD) re-fetched through L2 misses — far below the 256 MiB Infinity Cache, so served
                # on-die: an upper estimate of the share that never reaches HBM.
                q_in = {"k_extend": cst["ext_rays"] * 32, "k_shadow": cst["shadow_rays"] * (48 + 16), "k_mis": mis_rays * 32}.get(k)
                q_out = {"k_extend": cst["ext_rays"] * (16 + 4), "k_shadow": cst["shadow_rays"] * 16, "k_mis": mis_rays * 1}.get(k)
                if q_in:
                    e["hbm_counter_vs_queue_records"] = {
                        "queue_bytes_in_per_step": int(q_in), "queue_bytes_out_per_step": int(q_out),
                        "counter_over_queue": round(tr["hbm_bytes_per_step"] / (q_in + q_out), 3),
                        "infinity_cache_served_estimate_bytes": int(max(0, tr["hbm_read_bytes_per_step"] - q_in)),
                        "note": "queue records the kernel has to stream (rays / NEE records in — k_shadow: + the 16 B read of L —, hits / "
                                "results out); reads beyond its input records are scene data missing the 4 MiB L2 and found in the 256 MiB "
                                "Infinity Cache (FETCH_SIZE counts those too): an upper estimate of what is not HBM traffic"}
                e["hbm_counter_bytes_per_step"] = tr["hbm_bytes_per_step"]
                e["hbm_counter_gbs"] = round(tr["hbm_bytes_per_step"] / ms_step / 1e6, 1)
                # ONE definition for every kernel: counted memory-side bytes / the kernel's time alone on the GPU / 8 TB/s
                e["frac"] = round(e["hbm_counter_gbs"] / HBM_PEAK_GBS, 4)
                step_counter_bytes += tr["hbm_bytes_per_step"]
                step_counter_kernels.append(k)
                if k == "k_film" and "k_film_resolve" in pmc_traffic:
                    step_counter_bytes += pmc_traffic["k_film_resolve"]["hbm_bytes_per_step"]
            la = pmc_lanes.get(k)
            if la:
                # VALU issue: wave-instructions x 64 lane slots against CUs x SIMDs x 32 lanes x clock, over the kernel's time alone
                tl = la["SQ_INSTS_VALU"] * 64 / (ms_step * 1e-3) / 1e12
                e["valu"] = {"issued_tlaneops": round(tl, 2), "peak_tlaneops": round(VALU_PEAK_TLANEOPS, 1),
                             "issue_frac": round(tl / VALU_PEAK_TLANEOPS, 4), "lane_util": la.get("lane_util"),
                             "valu_busy": la.get("valu_busy"), "insts_valu_per_step": la["SQ_INSTS_VALU"],
                             # where a resident wave's time goes (SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES)
                             "wave_wait_any_frac": la.get("wave_wait_any_frac"), "wave_wait_inst_frac": la.get("wave_wait_inst_frac"),
                             "wave_active_inst_frac": la.get("wave_active_inst_frac"),
                             "counters_schedule": "the counter passes ran the two-stream command: the wave_* fractions describe the kernel beside its "
                                                  "neighbours of the other stream (a kernel queued behind another stream reads as waiting)",
                             "calibration": "profiles/r03_valu_calib.json: independent v_fma_f32 saturate at 0.5 wave-instructions per cycle per "
                                            "SIMD (= peak_tlaneops) from ~4 ready waves per SIMD, one wave alone issues 0.19-0.25; `valu_busy` "
                                            "(4 x SQ_ACTIVE_INST_VALU / SIMDs / busy cycles) reads 0.76 for one wave per SIMD and 1.5-1.8 when "
                                            "saturated, so it is not a utilisation out of 1; FP64 / packed FP32 cost 2 issue slots, "
                                            "transcendentals 4, a correctly rounded a/b ~17, sqrt ~22 — `issue_frac` counts every instruction as one slot"}
                if "TCC_REQ_sum" in la:
                    l2 = la["TCC_REQ_sum"] * 128 / (ms_step * 1e-3) / 1e9
                    e["l2"] = {"requests_per_step": la["TCC_REQ_sum"], "hit_rate": la.get("l2_hit_rate"),
                               "gbs_at_128B_per_request": round(l2, 1), "frac_of_l2_peak": round(l2 / L2_PEAK_GBS, 4)}
                if "TCP_TOTAL_CACHE_ACCESSES_sum" in la and vmem_peak:
                    # the vector-memory roofline: L1 accesses (one per lane and load instruction when the lanes diverge) per ns per
                    # CU against the ceiling measured for the same access shape (scene-sized table: 18 MB for the room, 1 MB when the
                    # tree fits an XCD's L2 many times over)
                    rate = la["TCP_TOTAL_CACHE_ACCESSES_sum"] / (ms_step * 1e6) / 256.0
                    peak = vmem_peak.get(18 if args.workload.startswith("boxroom") else 1) or max(vmem_peak.values())
                    e["vmem"] = {"l1_accesses_per_step": la["TCP_TOTAL_CACHE_ACCESSES_sum"], "l1_accesses_per_ns_cu": round(rate, 3),
                                 "calibrated_peak_per_ns_cu": peak, "frac": round(rate / peak, 4), "calibration": vmem_src,
                                 "ta_busy": la.get("ta_busy"), "td_busy": la.get("td_busy")}
            per_kernel[k] = e
        # the dominant kernel: the largest time alone on the GPU
        dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"])
        ms_sum, n_launch, bytes_all, units_k, note_k = fam[dom]
        launches_per_step = n_launch / steps
        ms_dom_step = alone_ms(dom) if have_alone else ms_sum / steps
        avg_ms = ms_dom_step / max(launches_per_step, 1)
        achieved = (bytes_all / launches_per_step) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        if dom in pmc_traffic:
            traffic = int(pmc_traffic[dom]["hbm_bytes_per_step"] / max(pmc_traffic[dom]["launches_in_step"], 1))
        frac_alg = achieved / HBM_PEAK_GBS
        frac_traffic = round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None
        roof = {
            "kernel": dom,
            "bound": BINDS.get(dom, "hbm"),
            "bound_note": BIND_NOTES[BINDS.get(dom, "hbm")],
            "priced_against": "hbm",
            "schedule": per_kernel[dom]["schedule"],
            # the contract's pair: algorithmic bytes per launch / launch time, against the HBM peak ...
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            # ... and the fraction, which for EVERY kernel is counted memory-side bytes per launch / launch time / peak
            # (`traffic` / avg_launch_ms / 8 TB/s; null without a committed counter set for this workload)
            "frac": frac_traffic,
            "frac_definition": "traffic / avg_launch_ms / peak: PMC-counted memory-side bytes (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc "
                               "passes of this command, profiles/" + str(tr_src) + ") over the kernel's launch time ALONE on the GPU (HIP events of the "
                               "one-stream steps of this run), the same definition for every kernel of roofline_all_kernels",
            "frac_algorithmic": round(frac_alg, 4),
            "frac_algorithmic_note": "achieved / peak with SURVEY.md 8(d)'s bytes (32 B per BVH node of the REFERENCE's layout + 48 B per triangle test "
                                     "+ 48 B per ray). It can exceed 1 for the traversal kernels: the scene (7 MB; the room 35 MB) is served from L2 / "
                                     "Infinity Cache, and a four-wide step does several of the reference's node visits per fetch — the bytes are a model "
                                     "of the reference's work, not traffic",
            "traffic": traffic,
            "traffic_source": tr_src,
            "launches_per_step": launches_per_step,
            "avg_launch_ms": round(avg_ms, 4),
            "algorithmic_bytes_per_launch": int(bytes_all / max(launches_per_step, 1)),
            "algorithmic_bytes_note": note_k,
            "units_per_launch": int(units_k / max(launches_per_step, 1)),
            "valu": per_kernel[dom].get("valu"),
            "l2": per_kernel[dom].get("l2"),
            "vmem": per_kernel[dom].get("vmem"),
            "lanes_source": la_src,
            "kernel_choice": "largest HIP-event time per step over all kernels, every kernel alone on the GPU" if have_alone
                             else "largest HIP-event time over all kernels of the (two-stream) step: this run had no one-stream steps",
        }
        if have_alone and not one_stream_timed:
            # the same kernel as the timed two-stream steps saw it (sharing the chip): what a rocprofv3 trace of the DEFAULT command shows
            o_ms = (ms_sum / steps) / max(launches_per_step, 1)
            o_ach = (bytes_all / launches_per_step) / (o_ms * 1e-3) / 1e9 if o_ms > 0 else 0.0
            roof["overlapped"] = {"avg_launch_ms": round(o_ms, 4), "achieved": round(o_ach, 1), "frac_algorithmic": round(o_ach / HBM_PEAK_GBS, 4),
                                  "frac": round(traffic / (o_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                                  "note": "In the timed steps the shadow / MIS kernels of a bounce run on a second stream beside the next bounce's "
                                          "k_extend and k_shade (each fills the other's tail): these HIP-event durations — the ones a rocprofv3 "
                                          "trace of the default command shows — include time spent sharing the GPU. Kept for comparison with such a "
                                          "trace; profiles/*_one_stream_kernel_stats.csv is the trace of `--schedule one-stream`, which agrees with "
                                          "the figures above"}
        copy_gbs = measured_copy_gbs(torch)
        roof["peak_measured_copy_gbs"] = round(copy_gbs, 1)
        # the whole step against HBM: counted bytes of every kernel of a step / ms_per_step / peak; and SURVEY.md 8(d)'s own
        # figure for the job: rays/s x B_ray / peak
        step_hbm_frac = round(step_counter_bytes / (primary["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if step_counter_bytes else None
        job_algorithmic_frac = round(mray * 1e6 * b_ray / 1e9 / HBM_PEAK_GBS, 4)
        out = {
            "metric": f"Mray/s on {workload_name} 1080p (path integrator, rays = Scene::Intersect + IntersectP calls)",
            "value": round(mray, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(primary["ms_per_step"], 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("scenes/killeroo-simple.pbrt (the reference's shipped scene file)" if args.workload == "killeroo"
                     else "synthetic scene generated by tests/boxroom.py (seed 12111)") + f"; {'Sobol' if args.sampler == 'sobol' else 'Halton'} samples generated on device",
            "config": {
                "workload": f"{workload_name} {args.xres}x{args.yres}, {total_spp} spp in all"
                            + (f" ({total_spp // world} spp-equivalents of work per GPU)" if world > 1 else "")
                            + f", path maxdepth 5, {args.sampler or 'halton'}, box filter, 16x16 tiles dealt diagonally over {world} rank(s)"
                            + (", films merged by one RCCL reduce (iile_dist_film_reduce)" if world > 1 else ""),
                "xres": args.xres, "yres": args.yres, "spp_total": total_spp,
                "baseline_config": ("2 (1080p x 64 spp, 1 GPU)" if (world, total_spp) == (1, 64) else
                                    "3 (1080p x 1024 spp over 8 GPUs)" if (world, total_spp) == (8, 1024) else
                                    "3's frame (1080p x 1024 spp) on fewer GPUs" if total_spp == 1024 else "scaled between 2 and 3"),
                "passes_per_step": primary["n_passes"],
            },
            "timed_film_verified": "bitwise equal to the instrumented step's film (all ranks)",
            "msamples_per_s": round(cam * steps / primary["elapsed"] / 1e6, 3),
            "rays_per_step": primary["rays_traced_step"],
            "reference_rays_per_step": rays_step,
            "value_reference_ray_equivalents": round(primary["mray_reference"], 2),
            "rays_note": "value counts the rays the timed kernels traced. The reference makes `reference_rays_per_step` Scene::Intersect / "
                         "IntersectP calls for this frame (counted by the instrumented step); the difference is EstimateDirect's BSDF-sampled "
                         "rays that the shade kernel proves unable to end on the sampled light, and the rays of bounce maxDepth, whose "
                         "intersection the path loop only uses to add emitted light after a specular bounce or from an infinite light — "
                         "neither exists in this scene (both exact: the film is bit for bit the same). `value_reference_ray_equivalents` "
                         "divides the reference's count by the same time.",
            "camera_samples_per_step": cam,
            "rays_per_camera_sample": round(primary["rays_traced_step"] / max(cam, 1), 4),
            "n_node_per_ray": round(n_node, 3),
            "n_tri_per_ray": round(n_tri, 4),
            "b_ray_bytes": round(b_ray, 1),
            "job_algorithmic_gbs": round(mray * 1e6 * b_ray / 1e9, 1),
            "schedule": args.schedule,
            "kernel_ms_per_step_overlapped_rank0": {k: round(agg[k] / steps, 3) for k in
                                                    ("ms_generate", "ms_extend", "ms_shade", "ms_shadow", "ms_mis", "ms_resolve", "ms_film", "ms_total")},
            "kernel_ms_per_step_one_stream": ({k: round(primary["alone"][k] / primary["n_alone"], 3) for k in
                                               ("ms_generate", "ms_extend", "ms_shade", "ms_shadow", "ms_mis", "ms_resolve", "ms_film", "ms_total")}
                                              if primary["n_alone"] else None),
            "roofline": roof,
            "roofline_all_kernels": per_kernel,
            "step_hbm_frac": step_hbm_frac,
            "step_hbm_frac_note": ("counted memory-side bytes of " + ", ".join(step_counter_kernels) + " over one step (profiles/" + str(tr_src) +
                                   ") / ms_per_step / 8 TB/s" if step_counter_bytes else "no committed counter set for this workload"),
            "job_algorithmic_frac": job_algorithmic_frac,
            "job_algorithmic_frac_note": "value x b_ray_bytes / 8 TB/s: SURVEY.md 8(d)'s `roofline.achieved` for the whole job (rays traced per second x "
                                         "the reference layout's bytes per ray), not traffic",
            "built": ge.build_provenance(compiled_now),
            "per_rank": dict(primary["per_rank"], **{
                "ranks": world,
                "rccl_ranks": (comm.ranks_seen if comm is not None else 1),   # ncclCommCount of the film-merge communicator
                "ms_render_min_mean_max": [round(min(primary["per_rank"]["ms_render"]), 3),
                                           round(sum(primary["per_rank"]["ms_render"]) / len(primary["per_rank"]["ms_render"]), 3),
                                           round(max(primary["per_rank"]["ms_render"]), 3)],
                "ms_reduce_max": round(max(primary["per_rank"]["ms_reduce"]), 3),
                "imbalance_max_over_mean": round(max(primary["per_rank"]["ms_render"]) /
                                                 max(sum(primary["per_rank"]["ms_render"]) / len(primary["per_rank"]["ms_render"]), 1e-9), 4),
                "note": "per timed step and rank, from events on the render stream: ms_render = iile_render of the rank's tiles, ms_reduce = "
                        "its part of the one film reduction (iile_dist_film_reduce; 0 at one rank). A rank that fails raises and exits "
                        "non-zero: the launcher (torch.distributed.run, max_restarts 0) tears the job down; nothing restarts a process "
                        "that has touched the GPU"}),
        }
        if other is not None:
            out["other_mode"] = {"scaling": other_name, "spp_total": other["total_spp"], "value": round(other["mray"], 2), "unit": "Mray/s",
                                 "ms_per_step": round(other["ms_per_step"], 3), "steps": other["steps"],
                                 "msamples_per_s": round(other["cam"] * other["steps"] / other["elapsed"] / 1e6, 3),
                                 "timed_film_verified": "bitwise equal to the instrumented step's film (all ranks)"}
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.scene, args.xres, args.yres, args.cpu_seconds, workload_name)
            out["speedup_vs_cpu_baseline"] = round(mray / max(out["cpu_baseline"]["value"], 1e-9), 1)
        if os.environ.get("IILE_GPU_LIB"):
            out["gpu_lib_override"] = os.environ["IILE_GPU_LIB"]
    if dist is not None:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
