"""The IISPT integrator's frame, end to end on the GPU (BASELINE config 5): the indirect pass (below), the
direct pass (IisptRenderRunner::run_direct -> iile_render_direct) and the final merge of the two film monitors
(IisptFilmMonitor::merge_into, src/integrators/iisptfilmmonitor.cpp:231-275, as src/integrators/iispt.cpp:405-446 ends).

IisptRenderRunner::run (src/integrators/iisptrenderrunner.cpp:216-596) per task of the schedule
(IisptScheduleMonitor::next_task, src/integrators/iisptschedulemonitor.cpp:40-79: square tasks of NUMBER_TILES = 10
tiles of `radius` pixels, the radius shrinking by sqrt(0.795...) after every sweep of the frame):

    hemi points of the task  --iile_iispt_hemi_points-->  probe cameras
    probe pass               --iile_render_probes------>  intensity / normals / distance images (HBM)
    normalizeMapsDownstream, IISPTNet, transformMapsUpstream  --iile_iispt_net_predict-->  one image per hemi point (HBM)
    per-pixel gather         --iile_iispt_gather------->  {f_beta * L, weight} per pixel (HBM)
    IisptFilmMonitor::add_n_samples (src/integrators/iisptfilmmonitor.cpp:47-72)  --iile_iispt_film_add-->  double sums per pixel

The same frame from C++: csrc/host/gpu_iispt_integrator.h (`iile_pbrt --integrator iispt`); tests/test_iispt_host.py holds the two
hosts' images against each other bit for bit.

Nothing but the hemi points' positions (a few KB per task) crosses PCIe. No trained weights ship with the reference:
with the default random-initialised network the numbers mean nothing; a checkpoint of the reference's ml/ training
loads into IISPTNet unchanged.
"""
import math

import numpy as np
import torch

NUMBER_TILES = 10  # iisptschedulemonitor.h:33


def schedule(bounds, n_tasks, radius_start=100.0, update_multiplier=None):
    """IisptScheduleMonitor::next_task for task numbers 0 .. n_tasks - 1 over film bounds (x0, y0, x1, y1):
    yields (x0, y0, x1, y1, tilesize). current_radius and update_multiplier are floats there (iisptschedulemonitor.h:36-38)."""
    bx0, by0, bx1, by1 = bounds
    if update_multiplier is None:
        update_multiplier = np.sqrt(np.float32(0.79541357))   # std::sqrt(0.79541357f)
    radius, nextx, nexty = np.float32(radius_start), bx0, by0
    for _ in range(n_tasks):
        eff = max(1, int(math.floor(float(radius))))
        size = eff * NUMBER_TILES
        yield nextx, nexty, min(nextx + size, bx1), min(nexty + size, by1), eff
        nextx += size
        if nextx >= bx1:
            nextx = bx0
            nexty += size
        if nexty >= by1:
            nexty = by0
            radius = np.float32(radius * np.float32(update_multiplier))


DIRECT_SAMPLES = 16  # PbrtOptions.iileDirectSamples, pbrt.h:178


class IisptFrame:
    """The two film monitors of IISPTIntegrator::Render (indirect: accumulated over tasks; direct: run_direct) and their
    merge; everything stays in HBM."""

    def __init__(self, binding, gpu_scene, pipeline, rng_seed=0):
        self.b, self.gpu, self.pipe = binding, gpu_scene, pipeline
        h, w = gpu_scene.host.film_shape
        # IisptPixel (iisptpixel.h): r, g, b sums and the weight sum, doubles
        self.film = torch.zeros((h, w, 4), dtype=torch.float64, device="cuda")         # film_monitor_indirect
        self.film_direct = torch.zeros((h, w, 4), dtype=torch.float64, device="cuda")  # film_monitor_direct
        self.direct_passes = 0
        self.counter = 0       # sampler_pixel_counter.x of the (single) runner
        self.rng_seed = rng_seed
        self.stats = {"tasks": 0, "hemi_points": 0, "probes": 0, "pixels": 0}

    @torch.no_grad()
    def run_task(self, x0, y0, x1, y1, tilesize):
        task = self.b.IisptTask(x0, y0, x1, y1, tilesize, self.counter, self.rng_seed)
        nx, ny = task.grid()
        valid, pos, dr = self.gpu.iispt_hemi_points(task)
        sel = valid.reshape(-1) == 1
        nn = torch.zeros((ny * nx, 32, 32, 3), dtype=torch.float32, device="cuda")
        if sel.any():
            # the gather reads the network's own row order (ImageFilm: row 0 = top scanline)
            pred, _, _, _ = self.pipe(pos.reshape(-1, 3)[sel], dr.reshape(-1, 3)[sel], film_rows=True)
            nn[torch.from_numpy(sel).cuda()] = pred
        h, w = y1 - y0, x1 - x0
        out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        self.gpu.iispt_gather(task, valid, pos, dr, nn_device_ptr=nn.data_ptr(), out_device_ptr=out.data_ptr())
        torch.cuda.synchronize()
        self.film[y0:y1, x0:x1] += out.double()  # add_n_samples
        self.counter += nx * ny + w * h
        self.rng_seed += w * h
        self.stats["tasks"] += 1
        self.stats["hemi_points"] += nx * ny
        self.stats["probes"] += int(sel.sum())
        self.stats["pixels"] += w * h

    def run(self, n_tasks, radius_start=100.0):
        h, w = self.gpu.host.film_shape
        for t in schedule((0, 0, w, h), n_tasks, radius_start):
            self.run_task(*t)
        return self.image()

    @torch.no_grad()
    def run_batched(self, n_tasks, radius_start=100.0, max_probes=32768, timers=None, rank=0, nranks=1):
        """The same film as run(): a task's result depends only on its rectangle, its sampler counter and its seed, all of
        which the schedule fixes in advance — so the stages run task-major instead of interleaved: hemi points of a group of
        tasks, ONE probe pass and ONE network call over all their probes (the reference pays a pipe round trip per probe),
        then the gathers. Groups are cut at max_probes hemi points. timers: dict that receives seconds per stage.
        rank / nranks: this process renders the tasks whose number is rank modulo nranks (the reference's render threads draw
        tasks from one schedule monitor, iispt.cpp:386-427: which thread renders which task does not change the task); counters and
        seeds advance over ALL tasks, so a task is the same task whoever renders it, and the monitors of the ranks add up to the
        single-rank frame's (reduce_monitors)."""
        import time
        h, w = self.gpu.host.film_shape
        tasks = list(schedule((0, 0, w, h), n_tasks, radius_start))

        def tick(name, t0):
            if timers is not None:
                torch.cuda.synchronize()
                timers[name] = timers.get(name, 0.0) + time.time() - t0

        i = 0
        while i < len(tasks):
            group, n_pts, n_pix = [], 0, 0
            # (a group never spans two sweeps: inside one sweep no two tasks share a pixel, which the one-launch film update needs)
            while i < len(tasks) and (not group or (n_pts < max_probes and tasks[i][:2] != (0, 0))):   # (a sweep starts at the film's corner)
                x0, y0, x1, y1, ts = tasks[i]
                task = self.b.IisptTask(x0, y0, x1, y1, ts, self.counter, self.rng_seed)
                nx, ny = task.grid()
                if i % nranks == rank:
                    group.append(task)
                    n_pts += nx * ny
                    n_pix += (x1 - x0) * (y1 - y0)
                self.counter += nx * ny + (x1 - x0) * (y1 - y0)
                self.rng_seed += (x1 - x0) * (y1 - y0)
                i += 1
            if not group and i < len(tasks):
                continue   # (none of this stretch of the schedule is this rank's)
            if not group:
                break
            # every stage runs over the whole group at once (iile_iispt_*_batch: one set of launches for all its tasks)
            t0 = time.time()
            stream = torch.cuda.current_stream().cuda_stream   # ONE stream orders the whole indirect pass (include/iile_gpu.h)
            valid, pos, dr = self.gpu.iispt_hemi_points_batch(group, stream=stream)
            tick("hemi_points", t0)
            t0 = time.time()
            sel = valid == 1
            nn = torch.zeros((n_pts, 32, 32, 3), dtype=torch.float32, device="cuda")
            if sel.any():
                # every probe's prediction lands in its hemi point's image (iile_iispt_net_predict's slot index)
                slot = torch.from_numpy(np.flatnonzero(sel).astype(np.int32)).cuda()
                self.pipe(pos[sel], dr[sel], film_rows=True, pred_out=nn, slot=slot)
            tick("probes_and_network", t0)
            t0 = time.time()
            out = torch.empty((n_pix, 4), dtype=torch.float32, device="cuda")
            self.gpu.iispt_gather_batch(group, valid, pos, dr, nn_device_ptr=nn.data_ptr(), out_device_ptr=out.data_ptr(), stream=stream)
            # add_n_samples for every pixel of every task of the group in ONE launch (iile_iispt_film_add: tasks of one sweep do not
            # overlap, so no two of its threads meet on a film pixel)
            self.gpu.iispt_film_add(group, out.data_ptr(), self.film.data_ptr(), stream=stream)
            for task in group:
                self.stats["tasks"] += 1
                self.stats["pixels"] += (task.y1 - task.y0) * (task.x1 - task.x0)
            tick("gather", t0)
            self.stats["hemi_points"] += n_pts
            self.stats["probes"] += int(sel.sum())
        torch.cuda.synchronize()
        return self.image()

    def run_direct(self, n_passes=DIRECT_SAMPLES, rank=0, nranks=1):
        """IisptRenderRunner::run_direct: n_passes more passes of DirectProgressiveIntegrator::RenderOnePass into the direct
        monitor (iile_render_direct; pass numbers continue where the last call stopped). rank / nranks: this process renders
        the contiguous block [n rank / N, n (rank + 1) / N) of them (a pass is its number — seed 6284 + 17 p — whoever renders it:
        the reference's threads draw pass numbers from the schedule monitor, iisptrenderrunner.cpp:617-627)."""
        p0, p1 = (n_passes * rank) // nranks, (n_passes * (rank + 1)) // nranks
        if p1 > p0:
            self.gpu.render_direct(p1 - p0, first_pass=self.direct_passes + p0, film_device_ptr=self.film_direct.data_ptr(),
                                   accumulate=self.direct_passes > 0, stream=torch.cuda.current_stream().cuda_stream)
        elif self.direct_passes == 0:
            self.film_direct.zero_()
        self.direct_passes += n_passes
        return self.film_direct

    def reduce_monitors(self, dist=None, others=()):
        """The ranks' film monitors added up: IisptFilmMonitor::add_n_samples is a sum of doubles per pixel, so the monitors of
        ranks that rendered disjoint tasks and passes add to the frame's (an indirect pixel belongs to one task per sweep; a direct
        pixel's passes are float values summed in doubles — exact whatever the order, short of 2^29 between the largest and the
        smallest). dist: an initialised torch.distributed (one all-reduce per monitor, RCCL); others: frames of the other ranks
        in THIS process (what the tests and tools/iispt_shard_probe.py do on one GPU)."""
        for o in others:
            self.film += o.film
            self.film_direct += o.film_direct
            for k in self.stats:
                self.stats[k] += o.stats[k]
        if dist is not None:
            dist.all_reduce(self.film)
            dist.all_reduce(self.film_direct)
        return self

    @staticmethod
    def _normalised(monitor):
        """IisptPixel::normalize: sums over the weight where it is positive (a pixel never written stays 0)."""
        wgt = monitor[..., 3:4]
        return torch.where(wgt > 0, monitor[..., :3] / torch.where(wgt > 0, wgt, torch.ones_like(wgt)), monitor[..., :3])

    def indirect_image(self):
        """film_monitor_indirect->to_intensity_film(): /tmp/iispt_indirect.exr of the reference."""
        return self._normalised(self.film).float()

    def direct_image(self):
        """film_monitor_direct->to_intensity_film(): /tmp/iispt_direct.exr of the reference."""
        return self._normalised(self.film_direct).float()

    def image(self):
        """The integrator's output (iispt.cpp:436-446): film_monitor_direct->merge_into(film_monitor_indirect) — both
        monitors normalised, added, weight 1 — through to_intensity_film: float RGB per pixel."""
        h, w = self.gpu.host.film_shape
        rgb = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        self.b.iispt_film_merge(self.film_direct.data_ptr(), self.film.data_ptr(), h * w, rgb.data_ptr())
        return rgb
