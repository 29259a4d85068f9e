// gpu_iispt_integrator.h — the IISPT integrator's frame for C++ hosts (BASELINE config 5), written against the
// C ABI only (include/iile_host.h, include/iile_gpu.h).
//
//   reference                                                        here
//   IISPTIntegrator(maxDepth, camera, pixelBounds, dcamera,          iile::GpuIisptIntegrator
//                   sampler, rrThreshold, lightStrategy)
//     (src/integrators/iispt.h, iispt.cpp:790-820)
//   IISPTIntegrator::render_normal_2 (iispt.cpp:357-446)             GpuIisptIntegrator::Render
//   IisptScheduleMonitor::next_task                                  iile::IisptSchedule::Next
//     (src/integrators/iisptschedulemonitor.cpp:9-79)
//   PbrtOptions.iileIndirectTasks / iileDirectSamples / iisptHemiSize IisptOptions (--iileIndirect= / --iileDirect= / --iispt_hemi_size=,
//     (src/core/pbrt.h:176-179, src/main/pbrt.cpp:167-178)            same defaults)
//   the Python child per thread that runs IISPTNet                   iile_iispt_net_load + iile_iispt_net_predict: the network is
//     (src/integrators/iisptnnconnector.cpp, ml/main_stdio_net.py)     a device object of this process; its weights come from a flat
//                                                                      file (binding.save_net_weights writes one from a checkpoint)
//
// render_normal_2 runs one IisptRenderRunner per CPU thread: every runner takes tasks from the schedule monitor until task number
// iileIndirectTasks, and direct passes until pass number iileDirectSamples; both film monitors are shared. A task's result depends on
// its rectangle, its sampler counter and its RNG only, so the stages run task-major here: hemi points of all tasks of a sweep (cut at
// max_probes hemi points), ONE probe pass and ONE network call over their probes, the gathers, one film update. Everything between the
// hemi points' positions (a few KB per task, read back to place the probe cameras) and the final image stays in HBM.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <chrono>

#include "gpu_integrator.h"

namespace iile {

struct IisptOptions {
    int indirect_tasks = 16;   // PbrtOptions.iileIndirectTasks
    int direct_samples = 16;   // PbrtOptions.iileDirectSamples
    int hemi_size = 32;        // PbrtOptions.iisptHemiSize (the network's input side: 32 is the only one it is trained for)
    std::string net_file;      // --iisptNet= / $IILE_IISPT_NET: IILENET1 weights (iile_iispt_net_load)
    int max_probes = 32768;    // hemi points per group of tasks (one set of probe-pass / gather launches each)
    int net_batch = 8192;      // probes per set of network launches: 1.19 MiB of activations each; no faster beyond (iile_iispt_net_predict halves it if the device is short of memory)
    std::string indirect_out, direct_out;   // /tmp/iispt_indirect.exr, /tmp/iispt_direct.exr of the reference; empty: not written
    // The frame over several GPUs, one process each (iile_pbrt --gpurank r/N): this process renders the tasks whose number is rank modulo
    // nranks and the block [n rank / N, n (rank + 1) / N) of the direct passes — what the reference's threads draw from ONE schedule
    // monitor (iispt.cpp:386-427) is the same task or pass whoever renders it — and the two film monitors are summed to rank 0
    // (iile_dist_monitor_reduce) before the merge. comm == nullptr with nranks > 1: the share alone (its images cover its tasks: tests).
    int rank = 0, nranks = 1;
    iile_dist *comm = nullptr;
};

struct IisptScheduleTask {
    int x0, y0, x1, y1, tilesize, pass, taskNumber;
};

// IisptScheduleMonitor (iisptschedulemonitor.cpp:9-79), the same environment variables, float arithmetic as there
class IisptSchedule {
  public:
    IisptSchedule(int x0, int y0, int x1, int y1) : bx0_(x0), by0_(y0), bx1_(x1), by1_(y1), nextx_(x0), nexty_(y0) {
        const char *e = std::getenv("IISPT_SCHEDULE_RADIUS_START");
        radius_ = e ? std::strtof(e, nullptr) : 100.0f;
        e = std::getenv("IISPT_SCHEDULE_RADIUS_RATIO");
        mult_ = e ? std::strtof(e, nullptr) : std::sqrt(0.79541357f);
    }
    IisptScheduleTask Next() {
        int eff = int(std::floor(radius_));
        if (eff < 1) eff = 1;
        const int size = eff * kNumberTiles;
        IisptScheduleTask t = {nextx_, nexty_, std::min(nextx_ + size, bx1_), std::min(nexty_ + size, by1_), eff, pass_, task_++};
        nextx_ += size;
        if (nextx_ >= bx1_) {
            nextx_ = bx0_;
            nexty_ += size;
        }
        if (nexty_ >= by1_) {
            nexty_ = by0_;
            radius_ *= mult_;
            ++pass_;
        }
        return t;
    }

  private:
    static const int kNumberTiles = 10;   // iisptschedulemonitor.h:33
    int bx0_, by0_, bx1_, by1_, nextx_, nexty_;
    float radius_, mult_;
    int pass_ = 0, task_ = 0;
};

class GpuIisptIntegrator : public Integrator {
  public:
    GpuIisptIntegrator(std::string output, IisptOptions opt) : output_(std::move(output)), opt_(std::move(opt)) {}

    struct Stats {
        int tasks = 0;
        long long hemi_points = 0, probes = 0, pixels = 0;
    } stats;

    // One process, several devices (iile_pbrt --integrator iispt --gpus N): a host thread per device, each a GpuIisptIntegrator of its own with
    // rank r of n and a communicator made in the thread — the frame's shares and the monitor reduction of the one-process-per-GPU form —
    // under the path integrator's rules for leaving together (device_gang.h: votes with deadlines before anybody enters a collective).
    bool RenderAllDevices(const Scene &scene, int n) {
        const int visible = iile_device_count();
        double timeout_s = 120.0;
        if (const char *e = std::getenv("IILE_DIST_TIMEOUT_S")) timeout_s = std::atof(e) > 0 ? std::atof(e) : timeout_s;
        uint8_t id[IILE_DIST_ID_BYTES];
        if (visible >= 1 && n >= 1 && n <= visible && iile_dist_unique_id(id) != IILE_OK) {
            fprintf(stderr, "Error: multi-GPU set-up: %s\n", iile_dist_last_error());
            return false;
        }
        struct Backend {
            GpuIisptIntegrator *self;
            const Scene &scene;
            const uint8_t *id;
            double timeout_s;
            int n;
            std::vector<Stats> stats;
            int fault_rank = -1;
            std::string fault;   // $IILE_DEBUG_GANG_FAULT = select:R | create:R (tests)
            bool SelectDevice(int r) {
                if ((fault == "select" && r == fault_rank) || iile_device_select(r) != IILE_OK) {
                    fprintf(stderr, "Error: IISPT: device %d: %s\n", r, fault == "select" && r == fault_rank ? "injected fault" : iile_last_error());
                    return false;
                }
                return true;
            }
            void *CreateComm(int r, int nn) {
                iile_dist *comm = nullptr;
                if ((fault == "create" && r == fault_rank) || iile_dist_create_deadline(id, r, nn, timeout_s, &comm) != IILE_OK) {
                    fprintf(stderr, "Error: multi-GPU set-up (device %d of %d): %s\n", r, nn, fault == "create" && r == fault_rank ? "injected fault" : iile_dist_last_error());
                    return nullptr;
                }
                return comm;
            }
            bool Run(int r, void *comm) {
                IisptOptions o = self->opt_;
                o.rank = r, o.nranks = n, o.comm = static_cast<iile_dist *>(comm);
                GpuIisptIntegrator part(self->output_, o);
                const bool ok = part.Render(scene);
                stats[size_t(r)] = part.stats;
                return ok;
            }
            void DestroyComm(void *comm) { iile_dist_destroy(static_cast<iile_dist *>(comm)); }
            void AbortComm(void *comm) { iile_dist_abort(static_cast<iile_dist *>(comm)); }
        } be{this, scene, id, timeout_s, n, std::vector<Stats>(size_t(n > 0 ? n : 0)), -1, std::string()};
        if (const char *f = std::getenv("IILE_DEBUG_GANG_FAULT")) {
            const std::string spec(f);
            const size_t colon = spec.find(':');
            if (colon != std::string::npos) be.fault = spec.substr(0, colon), be.fault_rank = std::atoi(spec.c_str() + colon + 1);
        }
        std::string why;
        if (!RunGang(be, n, visible, timeout_s, &why)) {
            fprintf(stderr, "Error: IISPT: the frame was given up on every device: %s\n", why.c_str());
            return false;
        }
        stats = be.stats[0];   // (Render sums the ranks' statistics)
        return true;
    }

    bool Render(const Scene &scene) override {
        if (!scene.ok()) return false;
        const bool timing = std::getenv("IILE_TIMING") != nullptr;
        const auto t_start = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            if (stream_) (void)iile_stream_wait(stream_);
            fprintf(stderr, "iile_pbrt timing:   + %8.3f s  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), what);
        };
        const iile_film_desc *f = scene.film();
        if (scene.desc()->probe.hemi_size != 32 || opt_.hemi_size != 32) return Fail("the IISPT network takes 32 x 32 probes (--iispt_hemi_size=32)");
        if (f->crop_x0 != 0 || f->crop_y0 != 0 || f->crop_x1 != f->xres || f->crop_y1 != f->yres)
            return Fail("the IISPT integrator renders the whole film (no cropwindow)");
        if (opt_.net_file.empty()) return Fail("the IISPT integrator needs the network's weights: --iisptNet=<file> or $IILE_IISPT_NET");
        const int w = f->xres, h = f->yres;
        const size_t n_pix = size_t(w) * size_t(h);
        bool ok = true;
        if (iile_scene_create(scene.desc(), &gpu_) != IILE_OK || iile_iispt_net_load(opt_.net_file.c_str(), &net_) != IILE_OK ||
            iile_stream_create(&stream_) != IILE_OK)   // ONE stream orders the frame: no stage relies on the null stream's implicit order
            ok = Fail(iile_last_error());
        double *film_indirect = nullptr, *film_direct = nullptr;
        float *rgb_dev = nullptr;
        ok = ok && Alloc(reinterpret_cast<void **>(&film_indirect), n_pix * 4 * sizeof(double)) &&
             Alloc(reinterpret_cast<void **>(&film_direct), n_pix * 4 * sizeof(double)) && Alloc(reinterpret_cast<void **>(&rgb_dev), n_pix * 3 * sizeof(float)) &&
             Check(iile_device_zero(film_indirect, n_pix * 4 * sizeof(double), stream_));
        lap("device initialised, scene uploaded, network loaded, films allocated");
        // ---- the indirect pass: IisptRenderRunner::run for task numbers 0 .. iileIndirectTasks - 1
        if (ok) {
            // camera->film->GetSampleBounds(): the film's pixels under the box filter of radius 0.5 (pbrt's default; BASELINE's scene).
            // Under a wider filter the reference's monitors — and its output image — grow by the filter's margin; here the frame
            // stays the film's pixels (what iile_render_direct and Film::WriteImage cover).
            if (f->samp_x0 != 0 || f->samp_y0 != 0 || f->samp_x1 != w || f->samp_y1 != h)
                fprintf(stderr, "Warning: IISPT: the pixel filter is wider than one pixel; the frame covers the film's %d x %d pixels, not the sample bounds\n", w, h);
            IisptSchedule schedule(0, 0, w, h);
            uint32_t counter = 0;   // sampler_pixel_counter.x of the runner (iisptrenderrunner.cpp:941-953)
            uint64_t seed = 0;      // the runner's RNG: one stream per film pixel here (iile_iispt_task::rng_seed)
            std::vector<iile_iispt_task> group;
            long long n_pts = 0;
            int group_pass = 0;
            auto flush = [&]() {
                const bool done = group.empty() || RunGroup(group, film_indirect, w, h);
                group.clear();
                n_pts = 0;
                return done;
            };
            for (int k = 0; k < opt_.indirect_tasks && ok; ++k) {
                const IisptScheduleTask s = schedule.Next();
                iile_iispt_task t = {s.x0, s.y0, s.x1, s.y1, s.tilesize, counter, seed};
                // (a group never spans two sweeps: inside one sweep no two tasks share a pixel, which the one-launch film update needs)
                if (!group.empty() && (group_pass != s.pass || n_pts >= opt_.max_probes)) ok = flush();
                group_pass = s.pass;
                const long long pts = (long long)iile_iispt_grid_count(t.x0, t.x1, t.tilesize) * iile_iispt_grid_count(t.y0, t.y1, t.tilesize);
                const long long pix = (long long)(t.x1 - t.x0) * (t.y1 - t.y0);
                if (k % opt_.nranks == opt_.rank) {   // (counter and seed advance over every task: a task is the same task whoever renders it)
                    group.push_back(t);
                    n_pts += pts;
                }
                counter += uint32_t(pts + pix);
                seed += uint64_t(pix);
            }
            ok = ok && flush();
        }
        lap("indirect pass");
        // ---- the direct pass: IisptRenderRunner::run_direct, passes 0 .. iileDirectSamples - 1
        if (ok) {
            iile_direct_params dp = {};
            dp.first_pass = (opt_.direct_samples * opt_.rank) / opt_.nranks;
            dp.n_passes = (opt_.direct_samples * (opt_.rank + 1)) / opt_.nranks - dp.first_pass;
            dp.film_on_device = 1;
            dp.stream = stream_;
            if (dp.n_passes > 0)
                ok = Check(iile_render_direct(gpu_, &dp, film_direct));
            else
                ok = Check(iile_device_zero(film_direct, n_pix * 4 * sizeof(double), stream_));
        }
        lap("direct pass");
        // ---- several GPUs: every rank says whether its share is whole, then the two monitors are summed to rank 0
        if (opt_.comm) {
            int32_t all_ok = 0;
            if (iile_dist_all_ok(opt_.comm, ok ? 1 : 0, &all_ok) != IILE_OK) ok = Fail(iile_dist_last_error());
            ok = ok && all_ok != 0;
            if (ok && (iile_dist_monitor_reduce(opt_.comm, film_indirect, int64_t(n_pix) * 4, 0, stream_) != IILE_OK ||
                       iile_dist_monitor_reduce(opt_.comm, film_direct, int64_t(n_pix) * 4, 0, stream_) != IILE_OK ||
                       iile_dist_wait(opt_.comm, stream_) != IILE_OK))
                ok = Fail(iile_dist_last_error());
            uint64_t totals[4] = {uint64_t(stats.tasks), uint64_t(stats.hemi_points), uint64_t(stats.probes), uint64_t(stats.pixels)};
            if (ok && iile_dist_sum_u64(opt_.comm, totals, 4) != IILE_OK) ok = Fail(iile_dist_last_error());
            stats.tasks = int(totals[0]), stats.hemi_points = (long long)totals[1], stats.probes = (long long)totals[2], stats.pixels = (long long)totals[3];
            lap("monitors summed over the ranks");
        }
        const bool writes = !opt_.comm || opt_.rank == 0;   // (rank 0 holds the summed monitors)
        // ---- iispt.cpp:425-446: the two monitors as images, their merge as the frame
        std::vector<float> rgb(n_pix * 3);
        auto write = [&](const double *a, const double *b, const std::string &path) {
            if (path.empty()) return true;
            if (!Check(iile_iispt_film_merge(a, b, int64_t(n_pix), rgb_dev, stream_)) ||
                !Check(iile_device_download(rgb.data(), rgb_dev, n_pix * 3 * sizeof(float), stream_)))
                return false;
            if (iile_host_write_image(path.c_str(), f, rgb.data()) != 0) return Fail(iile_host_last_error());
            return true;
        };
        if (ok && writes && (!opt_.indirect_out.empty() || !opt_.direct_out.empty())) {
            // to_intensity_film of ONE monitor = the merge with an empty one (a pixel of weight 0 contributes its sums, zeros)
            double *zero = nullptr;
            ok = Alloc(reinterpret_cast<void **>(&zero), n_pix * 4 * sizeof(double)) && Check(iile_device_zero(zero, n_pix * 4 * sizeof(double), stream_)) &&
                 write(film_indirect, zero, opt_.indirect_out) && write(film_direct, zero, opt_.direct_out);
        }
        ok = ok && (!writes || write(film_direct, film_indirect, output_));
        lap("images written");
        if (stream_) (void)iile_stream_wait(stream_);
        for (void *p : allocs_) iile_device_free(p);
        allocs_.clear();
        for (Buffer *b : {&inten_, &nrm_, &dist_, &nn_, &slot_, &out_}) {
            if (b->p) iile_device_free(b->p);
            *b = Buffer();
        }
        if (net_) iile_iispt_net_destroy(net_);
        if (gpu_) iile_scene_destroy(gpu_);
        if (stream_) iile_stream_destroy(stream_);
        stream_ = nullptr;
        net_ = nullptr;
        gpu_ = nullptr;
        return ok;
    }

  private:
    struct Buffer {
        void *p = nullptr;
        size_t bytes = 0;
    };
    bool Fail(const char *msg) {
        fprintf(stderr, "Error: IISPT: %s\n", msg);
        return false;
    }
    bool Check(int rc) { return rc == IILE_OK ? true : Fail(iile_last_error()); }
    bool Alloc(void **p, size_t bytes) {
        if (!Check(iile_device_alloc(bytes, p))) return false;
        allocs_.push_back(*p);
        return true;
    }
    bool Reserve(Buffer &b, size_t bytes) {   // grown on demand, kept for the frame (the groups of a frame have similar sizes)
        if (bytes <= b.bytes) return true;
        if (b.p) iile_device_free(b.p);
        b = Buffer();
        if (!Check(iile_device_alloc(bytes + bytes / 4, &b.p))) return false;
        b.bytes = bytes + bytes / 4;
        return true;
    }
    // hemi points -> probe pass -> network -> gather -> add_n_samples for the tasks of one group (iisptrenderrunner.cpp:248-596)
    bool RunGroup(const std::vector<iile_iispt_task> &tasks, double *film, int w, int h) {
        const int n_tasks = int(tasks.size());
        size_t n_pts = 0, n_pix = 0;
        for (const iile_iispt_task &t : tasks) {
            n_pts += size_t(iile_iispt_grid_count(t.x0, t.x1, t.tilesize)) * size_t(iile_iispt_grid_count(t.y0, t.y1, t.tilesize));
            n_pix += size_t(t.x1 - t.x0) * size_t(t.y1 - t.y0);
        }
        valid_.resize(n_pts);
        pos_.resize(3 * n_pts);
        dir_.resize(3 * n_pts);
        if (!Check(iile_iispt_hemi_points_batch(gpu_, tasks.data(), n_tasks, valid_.data(), pos_.data(), dir_.data(), stream_))) return false;
        // the probes: one per hemi point that found a scattering surface, its image written to that hemi point's slot
        cpos_.clear();
        cdir_.clear();
        slots_.clear();
        for (size_t i = 0; i < n_pts; ++i)
            if (valid_[i] == 1) {
                slots_.push_back(int32_t(i));
                cpos_.insert(cpos_.end(), &pos_[3 * i], &pos_[3 * i] + 3);
                cdir_.insert(cdir_.end(), &dir_[3 * i], &dir_[3 * i] + 3);
            }
        const size_t n_probes = slots_.size();
        const size_t img = 32 * 32 * sizeof(float);
        if (!Reserve(nn_, n_pts * 3 * img) || !Reserve(out_, n_pix * 4 * sizeof(float)) || !Check(iile_device_zero(nn_.p, n_pts * 3 * img, stream_))) return false;
        if (n_probes) {
            if (!Reserve(inten_, n_probes * 3 * img) || !Reserve(nrm_, n_probes * 3 * img) || !Reserve(dist_, n_probes * img) ||
                !Reserve(slot_, n_probes * sizeof(int32_t)))
                return false;
            if (!Check(iile_render_probes(gpu_, int32_t(n_probes), cpos_.data(), cdir_.data(), static_cast<float *>(inten_.p), static_cast<float *>(nrm_.p),
                                          static_cast<float *>(dist_.p), 1, nullptr, stream_)) ||
                !Check(iile_device_upload(slot_.p, slots_.data(), n_probes * sizeof(int32_t), stream_)) ||
                // (film_rows: the gather reads the network's own row order, ImageFilm's)
                !Check(iile_iispt_net_predict(net_, static_cast<float *>(inten_.p), static_cast<float *>(nrm_.p), static_cast<float *>(dist_.p),
                                              static_cast<float *>(nn_.p), static_cast<const int32_t *>(slot_.p), int32_t(n_probes), 1, opt_.net_batch, stream_)))
                return false;
        }
        if (!Check(iile_iispt_gather_batch(gpu_, tasks.data(), n_tasks, valid_.data(), pos_.data(), dir_.data(), static_cast<float *>(nn_.p), 1,
                                           static_cast<float *>(out_.p), 1, stream_)) ||
            !Check(iile_iispt_film_add(gpu_, tasks.data(), n_tasks, static_cast<float *>(out_.p), film, w, h, stream_)))
            return false;
        stats.tasks += n_tasks;
        stats.hemi_points += (long long)n_pts;
        stats.probes += (long long)n_probes;
        stats.pixels += (long long)n_pix;
        return true;
    }

    std::string output_;
    IisptOptions opt_;
    iile_scene *gpu_ = nullptr;
    iile_iispt_net *net_ = nullptr;
    void *stream_ = nullptr;   // the frame's one stream (iile_stream_create)
    std::vector<void *> allocs_;
    Buffer inten_, nrm_, dist_, nn_, slot_, out_;
    std::vector<uint8_t> valid_;
    std::vector<float> pos_, dir_, cpos_, cdir_;
    std::vector<int32_t> slots_;
};

// CreateIISPTIntegrator (iispt.cpp:790-820): "maxdepth" and the rest arrive through the scene file's Integrator line
inline GpuIisptIntegrator *CreateGpuIisptIntegrator(const ParamSet &, const std::string &output, const IisptOptions &opt) {
    return new GpuIisptIntegrator(output, opt);
}

}  // namespace iile
