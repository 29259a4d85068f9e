// iile_pbrt — command-line front end: `pbrt scene.pbrt` for the GPU path.
//
//   iile_pbrt scene.pbrt [--outfile out.pfm] [--xres N --yres N --spp N --maxdepth N] [--stats]
//
// Mirrors src/main/pbrt.cpp:97-219 (argument loop, ParseFile, Render) on top of
// the C ABI: libiile_host loads and flattens the scene, libiile_gpu renders it,
// the film is normalised and written as PFM.
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../host/gpu_integrator.h"

int main(int argc, char **argv) {
    std::string scene_file, out = "iile.pfm";
    iile::ParamSet ps;
    bool stats = false;
    for (int i = 1; i < argc; ++i) {
        auto arg_int = [&](int &dst) {
            if (i + 1 < argc) dst = atoi(argv[++i]);
        };
        if (!strcmp(argv[i], "--outfile") && i + 1 < argc)
            out = argv[++i];
        else if (!strcmp(argv[i], "--xres"))
            arg_int(ps.xresolution);
        else if (!strcmp(argv[i], "--yres"))
            arg_int(ps.yresolution);
        else if (!strcmp(argv[i], "--spp"))
            arg_int(ps.pixelsamples);
        else if (!strcmp(argv[i], "--maxdepth"))
            arg_int(ps.maxdepth);
        else if (!strcmp(argv[i], "--stats"))
            stats = true;
        else if (argv[i][0] == '-') {
            fprintf(stderr, "usage: iile_pbrt scene.pbrt [--outfile f.pfm] [--xres N] [--yres N] [--spp N] "
                            "[--maxdepth N] [--stats]\n");
            return 1;
        } else
            scene_file = argv[i];
    }
    if (scene_file.empty()) {
        fprintf(stderr, "iile_pbrt: no scene file given\n");
        return 1;
    }
    iile::Scene scene(scene_file, ps);
    std::unique_ptr<iile::GpuPathIntegrator> integrator(iile::CreateGpuPathIntegrator(ps, out, 0, 1, stats));
    if (!integrator->Render(scene)) return 1;
    const iile_stats &st = integrator->last_stats;
    printf("rendered %llu camera samples in %.1f ms -> %s\n", (unsigned long long)st.n_paths, st.ms_total, out.c_str());
    if (stats)
        printf("rays: %llu closest + %llu shadow; BVH nodes visited %llu + %llu; triangle tests %llu (%llu hits)\n",
               (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays,
               (unsigned long long)st.nodes_closest, (unsigned long long)st.nodes_any,
               (unsigned long long)st.tri_tests, (unsigned long long)st.tri_hits);
    return 0;
}
