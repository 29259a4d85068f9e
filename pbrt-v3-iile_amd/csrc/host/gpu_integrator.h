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
#include <string>
#include <vector>

#include "../../../include/iile_gpu.h"
#include "../../../include/iile_host.h"

namespace iile {

struct ParamSet {  // the two knobs CreatePathIntegrator reads; <= 0 keeps the scene file's value
    int maxdepth = 0;
    int xresolution = 0, yresolution = 0, pixelsamples = 0;
};

class Scene {
  public:
    explicit Scene(const std::string &pbrt_file, const ParamSet &ps = ParamSet()) {
        iile_host_overrides ov = {ps.xresolution, ps.yresolution, ps.pixelsamples, ps.maxdepth};
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
    // tile_rank / tile_nranks: this process renders tiles with index % nranks == rank
    GpuPathIntegrator(std::string output_pfm, int tile_rank = 0, int tile_nranks = 1, bool print_stats = false)
        : output_(std::move(output_pfm)), rank_(tile_rank), nranks_(tile_nranks), stats_(print_stats) {}

    bool Render(const Scene &scene) override {
        if (!scene.ok()) return false;
        iile_scene *gpu = nullptr;
        if (iile_scene_create(scene.desc(), &gpu) != IILE_OK) {
            fprintf(stderr, "Error: GPU path: %s\n", iile_last_error());
            return false;
        }
        const iile_film_desc *f = scene.film();
        const int w = f->crop_x1 - f->crop_x0, h = f->crop_y1 - f->crop_y0;
        std::vector<float> xyzw(size_t(4) * w * h), rgb(size_t(3) * w * h);
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
        iile_host_film_to_rgb(f, xyzw.data(), rgb.data());                 // Film::to_rgb_array
        if (!output_.empty() && iile_host_write_pfm(output_.c_str(), rgb.data(), w, h) != 0) {  // Film::WriteImage
            fprintf(stderr, "Error: %s\n", iile_host_last_error());
            return false;
        }
        return true;
    }
    iile_stats last_stats = {};

  private:
    std::string output_;
    int rank_, nranks_;
    bool stats_;
};

inline GpuPathIntegrator *CreateGpuPathIntegrator(const ParamSet &, const std::string &output_pfm, int tile_rank = 0,
                                                  int tile_nranks = 1, bool print_stats = false) {
    return new GpuPathIntegrator(output_pfm, tile_rank, tile_nranks, print_stats);
}

}  // namespace iile
