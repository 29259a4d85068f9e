// gpu_integrator.h — C++ host-side mirror of the reference's integrator surface
// for the GPU path, written against the C ABI only (include/iile_host.h,
// include/iile_gpu.h).
//
//   reference                                        here
//   class Integrator { virtual void Render(const Scene&) = 0; }   iile::Integrator
//     (src/core/integrator.h:53-58)
//   PathIntegrator(maxDepth, camera, sampler, pixelBounds,        iile::GpuPathIntegrator
//                  rrThreshold, lightSampleStrategy)
//     (src/integrators/path.h:50-60)
//   CreatePathIntegrator(const ParamSet&, sampler, camera)        iile::CreateGpuPathIntegrator
//     (src/integrators/path.h:70-72; parameters "maxdepth",
//      "rrthreshold" as in path.cpp:214-231)
//
// In the reference, camera/sampler/film objects carry the settings; here they
// live in the flattened iile_scene_desc, so `Scene` wraps the loaded scene and
// Render() is `SamplerIntegrator::Render` + `Film::WriteImage`
// (src/core/integrator.cpp:227-339). Errors are reported like the reference's
// Error() (src/core/error.h:54): a message on stderr, Render() returns false.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "device_gang.h"

#include "../../../include/iile_dist.h"
#include "../../../include/iile_gpu.h"
#include "../../../include/iile_host.h"

namespace iile {

struct ParamSet {  // the two knobs CreatePathIntegrator reads; <= 0 keeps the scene file's value
    int maxdepth = 0;
    int xresolution = 0, yresolution = 0, pixelsamples = 0;
    int sampler = IILE_SAMPLER_KEEP;  // IILE_SAMPLER_SOBOL: the fork's IILE_PATH_SAMPLES_OVERRIDE (path.cpp:202-212)
    int splitmethod = IILE_SPLIT_KEEP;  // BVHAccel "splitmethod" (bvh.cpp:740-760) in place of the file's
    bool bvh_on_device = false;         // split method "hlbvh" built by iile_bvh_build_hlbvh instead of the host builder
    bool quick = false;                 // pbrt --quick
};

class Scene {
  public:
    explicit Scene(const std::string &pbrt_file, const ParamSet &ps = ParamSet()) {
        iile_host_overrides ov = {ps.xresolution, ps.yresolution, ps.pixelsamples, ps.maxdepth, ps.sampler, ps.splitmethod, nullptr, ps.quick ? 1 : 0};
        if (ps.bvh_on_device)
            ov.bvh_build = [](int32_t n, const float *b6, int32_t maxp, iile_bvh_node *nodes, int32_t *n_nodes, int32_t *order, void *) {
                return iile_bvh_build_hlbvh(n, b6, maxp, nodes, n_nodes, order, nullptr);
            };
        if (iile_host_load_pbrt(pbrt_file.c_str(), &ov, &host_) != 0) {
            fprintf(stderr, "Error: %s\n", iile_host_last_error());
            host_ = nullptr;
        }
    }
    ~Scene() {
        if (host_) iile_host_scene_free(host_);
    }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    bool ok() const { return host_ != nullptr; }
    const iile_scene_desc *desc() const { return iile_host_scene_desc(host_); }
    const iile_film_desc *film() const { return iile_host_scene_film(host_); }
    int integrator() const {   // IILE_INTEGRATOR_*: the file's Integrator line (renderOptions->IntegratorName)
        iile_host_scene_info info = {};
        return host_ && iile_host_scene_get_info(host_, &info) == 0 ? info.integrator : IILE_INTEGRATOR_PATH;
    }
    std::string film_filename() const { return host_ ? iile_host_scene_film_filename(host_) : std::string(); }

  private:
    iile_host_scene *host_ = nullptr;
};

class Integrator {
  public:
    virtual ~Integrator() {}
    virtual bool Render(const Scene &scene) = 0;
};

class GpuPathIntegrator : public Integrator {
  public:
    // One process per GPU: with a communicator (iile_dist.h) this process renders the tiles iile_tile_owner gives
    // rank `iile_dist_rank(comm)` and the films are merged by one RCCL sum-reduction to rank 0, which writes the image
    // (the reference: tiles over threads, Film::MergeFilmTile under a mutex, src/core/film.cpp:135-148).
    // Without one: tile_rank / tile_nranks select a shard and the caller merges (tests), default all tiles.
    GpuPathIntegrator(std::string output_pfm, int tile_rank = 0, int tile_nranks = 1, bool print_stats = false, iile_dist *comm = nullptr)
        : output_(std::move(output_pfm)), rank_(comm ? iile_dist_rank(comm) : tile_rank), nranks_(comm ? iile_dist_size(comm) : tile_nranks),
          stats_(print_stats), comm_(comm) {}
    // One process, several devices: Render() of an integrator made without a communicator and without a shard uses
    // `devices` GPUs of this process (0 = every visible one; 1 GPU visible: the plain single-device path below).
    void UseDevices(int devices) { devices_ = devices; }

    bool Render(const Scene &scene) override {
        if (comm_) return RenderRanks(scene);
        if (nranks_ == 1 && devices_ >= 0) {
            const int visible = iile_device_count();
            const int n = devices_ == 0 ? visible : devices_;
            if (devices_ > 0 || visible > 1) return RenderAllDevices(scene, n);
        }
        if (!scene.ok()) return false;
        iile_scene *gpu = nullptr;
        if (iile_scene_create(scene.desc(), &gpu) != IILE_OK) {
            fprintf(stderr, "Error: GPU path: %s\n", iile_last_error());
            return false;
        }
        const iile_film_desc *f = scene.film();
        const size_t n_pix = size_t(f->crop_x1 - f->crop_x0) * size_t(f->crop_y1 - f->crop_y0);
        std::vector<float> xyzw(4 * n_pix);
        iile_render_params prm = {};
        prm.tile_rank = rank_;
        prm.tile_nranks = nranks_;
        prm.collect_stats = stats_ ? 1 : 0;
        iile_stats st;
        const int rc = iile_render(gpu, &prm, xyzw.data(), &st);
        iile_scene_destroy(gpu);
        if (rc != IILE_OK) {
            fprintf(stderr, "Error: GPU path: %s\n", iile_last_error());
            return false;
        }
        last_stats = st;
        return WriteFilm(f, xyzw);
    }
    iile_stats last_stats = {};

    // The reference enters integrator->Render(*scene) ONCE, in one process (src/core/api.cpp:1650-1662), and fans the tiles out
    // itself (ParallelFor2D over its thread pool, src/core/parallel.cpp:247-299). The same here over the GPUs of the process:
    // one host thread per device, each with its own iile_scene, the tiles dealt by iile_tile_owner, the films merged by the ONE
    // RCCL reduction of the N-process path — the threads rendezvous through an in-process unique id and then run exactly
    // RenderRanks. Rank 0's thread writes the image. n == 1 is the same code with a communicator of one rank (what the one-GPU
    // boxes of this pool can run: tests/test_gpu_parity.py).
    // Leaving together (device_gang.h): more devices asked for than visible is refused before a thread starts; a thread whose
    // device cannot be selected, or whose communicator set-up fails, votes no and NOBODY enters a collective; the communicator is
    // RCCL's non-blocking kind with a deadline (iile_dist_create_deadline; $IILE_DIST_TIMEOUT_S, default 120 s), so a rank that
    // stops answering later makes the others return an error instead of waiting in ncclReduce. Render() then returns false; the
    // process is never restarted or re-executed (it has touched the GPU).
    bool RenderAllDevices(const Scene &scene, int n) {
        const int visible = iile_device_count();
        double timeout_s = 120.0;
        if (const char *e = getenv("IILE_DIST_TIMEOUT_S")) timeout_s = atof(e) > 0 ? atof(e) : timeout_s;
        uint8_t id[IILE_DIST_ID_BYTES];
        if (visible >= 1 && n >= 1 && n <= visible && iile_dist_unique_id(id) != IILE_OK) {
            fprintf(stderr, "Error: multi-GPU set-up: %s\n", iile_dist_last_error());
            return false;
        }
        struct Backend {
            GpuPathIntegrator *self;
            const Scene &scene;
            const uint8_t *id;
            double timeout_s;
            std::vector<iile_stats> stats;
            int fault_rank = -1;
            std::string fault;   // $IILE_DEBUG_GANG_FAULT = select:R | create:R — fault injection for the tests (what a lost device looks like)
            bool SelectDevice(int r) {
                // (error strings are thread-local in both libraries: every thread reports its own)
                if ((fault == "select" && r == fault_rank) || iile_device_select(r) != IILE_OK) {
                    fprintf(stderr, "Error: GPU path: device %d: %s\n", r, fault == "select" && r == fault_rank ? "injected fault" : iile_last_error());
                    return false;
                }
                return true;
            }
            void *CreateComm(int r, int n) {
                iile_dist *comm = nullptr;
                if ((fault == "create" && r == fault_rank) || iile_dist_create_deadline(id, r, n, timeout_s, &comm) != IILE_OK) {
                    fprintf(stderr, "Error: multi-GPU set-up (device %d of %d): %s\n", r, n, fault == "create" && r == fault_rank ? "injected fault" : iile_dist_last_error());
                    return nullptr;
                }
                return comm;
            }
            bool Run(int r, void *comm) {
                GpuPathIntegrator part(self->output_, 0, 1, self->stats_, static_cast<iile_dist *>(comm));
                const bool ok = part.RenderRanks(scene);
                stats[size_t(r)] = part.last_stats;
                return ok;
            }
            void DestroyComm(void *comm) { iile_dist_destroy(static_cast<iile_dist *>(comm)); }
            void AbortComm(void *comm) { iile_dist_abort(static_cast<iile_dist *>(comm)); }
        } be{this, scene, id, timeout_s, std::vector<iile_stats>(size_t(n > 0 ? n : 0)), -1, std::string()};
        if (const char *f = getenv("IILE_DEBUG_GANG_FAULT")) {
            const std::string spec(f);
            const size_t colon = spec.find(':');
            if (colon != std::string::npos) be.fault = spec.substr(0, colon), be.fault_rank = atoi(spec.c_str() + colon + 1);
        }
        std::string why;
        if (!RunGang(be, n, visible, timeout_s, &why)) {
            fprintf(stderr, "Error: GPU path: the frame was given up on every device: %s\n", why.c_str());
            return false;
        }
        last_stats = be.stats[0];   // job totals (RenderRanks sums the counters over the ranks)
        return true;
    }

  private:
    bool WriteFilm(const iile_film_desc *f, const std::vector<float> &xyzw) {
        std::vector<float> rgb(xyzw.size() / 4 * 3);
        iile_host_film_to_rgb(f, xyzw.data(), rgb.data());                 // Film::to_rgb_array
        if (!output_.empty() && iile_host_write_image(output_.c_str(), f, rgb.data()) != 0) {  // Film::WriteImage: .exr or .pfm
            fprintf(stderr, "Error: %s\n", iile_host_last_error());
            return false;
        }
        return true;
    }
    // Every rank walks the same sequence of collectives whatever happens to it locally: a rank whose scene did not load,
    // whose allocation failed or whose render returned an error says so through iile_dist_all_ok and all ranks leave
    // together — nobody is left waiting inside ncclReduce for a rank that has already returned.
    bool Agree(bool ok) {
        int32_t all = 0;
        if (iile_dist_all_ok(comm_, ok ? 1 : 0, &all) != IILE_OK) {
            fprintf(stderr, "Error: multi-GPU status exchange: %s\n", iile_dist_last_error());
            return false;
        }
        if (ok && !all) fprintf(stderr, "Error: GPU path: rank %d stops because another rank failed\n", rank_);
        return all != 0;
    }
    bool RenderRanks(const Scene &scene) {
        iile_scene *gpu = nullptr;
        void *film_dev = nullptr;
        size_t n_pix = 0;
        const iile_film_desc *f = nullptr;
        bool ok = scene.ok();
        if (ok && iile_scene_create(scene.desc(), &gpu) != IILE_OK) {
            fprintf(stderr, "Error: GPU path: %s\n", iile_last_error());
            ok = false;
        }
        if (ok) {
            f = scene.film();
            n_pix = size_t(f->crop_x1 - f->crop_x0) * size_t(f->crop_y1 - f->crop_y0);
            // the film stays in HBM from the render through the merge; only rank 0 reads it back
            if (iile_device_alloc(4 * n_pix * sizeof(float), &film_dev) != IILE_OK) {
                fprintf(stderr, "Error: GPU path: %s\n", iile_last_error());
                ok = false;
            }
        }
        iile_stats st = {};
        if (ok) {
            iile_render_params prm = {};
            prm.tile_rank = rank_;
            prm.tile_nranks = nranks_;
            prm.collect_stats = stats_ ? 1 : 0;
            prm.film_on_device = 1;
            if (iile_render(gpu, &prm, static_cast<float *>(film_dev), &st) != IILE_OK) {
                fprintf(stderr, "Error: GPU path: %s\n", iile_last_error());
                ok = false;
            }
        }
        std::vector<float> xyzw;
        if (Agree(ok)) {  // all films are ready: the one collective of the frame (Film::MergeFilmTile, film.cpp:135-148)
            // (iile_dist_wait: the merge has finished, or the communicator's deadline has passed and this rank gives up)
            if (iile_dist_film_reduce(comm_, static_cast<float *>(film_dev), int64_t(n_pix), 0, nullptr) != IILE_OK ||
                iile_dist_wait(comm_, nullptr) != IILE_OK) {
                fprintf(stderr, "Error: film merge: %s\n", iile_dist_last_error());
                ok = false;
            }
            if (ok && rank_ == 0) {
                xyzw.resize(4 * n_pix);
                if (iile_device_download(xyzw.data(), film_dev, 4 * n_pix * sizeof(float), nullptr) != IILE_OK) {
                    fprintf(stderr, "Error: GPU path: %s\n", iile_last_error());
                    ok = false;
                }
            }
            // job totals of the counters (every rank takes part, with or without --stats on its own command line)
            uint64_t c[8] = {st.camera_rays, st.closest_rays, st.shadow_rays, st.nodes_closest, st.nodes_any, st.tri_tests, st.tri_hits, st.n_paths};
            double ms = st.ms_total;
            if (iile_dist_sum_u64(comm_, c, 8) == IILE_OK && iile_dist_max_f64(comm_, &ms, 1) == IILE_OK) {
                st.camera_rays = c[0], st.closest_rays = c[1], st.shadow_rays = c[2], st.nodes_closest = c[3];
                st.nodes_any = c[4], st.tri_tests = c[5], st.tri_hits = c[6], st.n_paths = c[7];
                st.ms_total = ms;
            } else {
                fprintf(stderr, "Error: job totals: %s\n", iile_dist_last_error());
                ok = false;
            }
            ok = Agree(ok);
        } else {
            ok = false;
        }
        if (film_dev) iile_device_free(film_dev);
        if (gpu) iile_scene_destroy(gpu);
        if (!ok) return false;
        last_stats = st;
        if (rank_ != 0) return true;  // rank 0 holds the merged film
        return WriteFilm(f, xyzw);
    }

  private:
    std::string output_;
    int rank_, nranks_;
    bool stats_;
    iile_dist *comm_;
    int devices_ = -1;   // >= 0: Render() may use several devices of this process (UseDevices)
};

inline GpuPathIntegrator *CreateGpuPathIntegrator(const ParamSet &, const std::string &output_pfm, int tile_rank = 0,
                                                  int tile_nranks = 1, bool print_stats = false, iile_dist *comm = nullptr) {
    return new GpuPathIntegrator(output_pfm, tile_rank, tile_nranks, print_stats, comm);
}

}  // namespace iile
