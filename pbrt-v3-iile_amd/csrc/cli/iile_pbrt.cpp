// iile_pbrt — command-line front end: `pbrt scene.pbrt` for the GPU path.
//
//   iile_pbrt scene.pbrt [--outfile out.exr|out.pfm] [--xres N --yres N --spp N --maxdepth N] [--stats] [--gpus N]
//             [--gpurank R/N --rendezvous FILE [--job TOKEN]]
//             [--integrator path|iispt] [--iisptNet=FILE] [--iileIndirect=TASKS] [--iileDirect=SAMPLES] [--iispt_hemi_size=32]
//
// Which integrator renders the frame is the scene file's Integrator line, as in MakeIntegrator (src/core/api.cpp:1720-1750): "path" ->
// GpuPathIntegrator, "iispt" -> GpuIisptIntegrator (csrc/host/gpu_iispt_integrator.h; --iileIndirect= / --iileDirect= /
// --iispt_hemi_size= keep the reference's spellings and defaults, src/main/pbrt.cpp:167-178). The reference starts one Python child per
// thread for the network (IISPT_STDIO_NET_PY_PATH); here --iisptNet=FILE (or $IILE_IISPT_NET) names the weights, an IILENET1 file as
// binding.save_net_weights writes from a checkpoint. --integrator overrides the file's line. The IISPT frame is a one-device job.
//
// One process, the whole node (default): with several GPUs visible the frame's tiles are dealt over all of them, one host
// thread per device, and merged by one RCCL reduction — `pbrt scene.pbrt` needs no launcher, as the reference's one Render()
// call fans out over its threads (src/core/api.cpp:1650-1662). --gpus N uses N of them (--gpus 1 runs the same code with a
// communicator of one rank).
// Multi-GPU: start N copies, one per GPU, with --gpurank 0/N .. N-1/N and a common --rendezvous file on a shared
// file system (rank 0 publishes the RCCL id there; --job TOKEN, any number the launcher picks per launch, ties the file
// to this launch; a rank that does not show up within $IILE_DIST_TIMEOUT_S — default 120 s — makes the others exit 1 instead of
// waiting for ever). Rank R uses GPU R modulo the visible devices, renders its tiles
// and the films are merged on rank 0 by one RCCL reduction (include/iile_dist.h); rank 0 writes the image.
// --gpurank 0/1 is the same code path with a communicator of one rank (tools/multi_gpu_cmdline.sh prints the N-rank
// command lines).
// The IISPT integrator over several GPUs is --gpurank too (or --gpus N: threads of one process): rank R renders the tasks whose number is
// R modulo N and its block of the direct passes, the two film monitors are summed on rank 0 (iile_dist_monitor_reduce), which merges and
// writes the images.
//
// Mirrors src/main/pbrt.cpp:97-219 (argument loop, ParseFile, Render) on top of
// the C ABI: libiile_host loads and flattens the scene, libiile_gpu renders it,
// the film is normalised and written as PFM.
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "../host/gpu_iispt_integrator.h"

int main(int argc, char **argv) {
    // $IILE_TIMING: wall time of the process's phases to stderr (where does a slow start come from?)
    const bool timing = getenv("IILE_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "iile_pbrt timing: %8.3f s  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), what);
    };
    std::string scene_file, out;  // --outfile, else the scene's Film "filename" (as pbrt: src/main/pbrt.cpp:137, film.cpp:262)
    iile::ParamSet ps;
    bool stats = false, quiet = false;
    int gpu_rank = 0, gpu_nranks = 1;
    int gpus = 0;   // --gpus N: devices of THIS process (0: all visible)
    bool gpus_given = false;
    bool ranked = false;  // --gpurank given: render through the communicator branch, also for N = 1
    unsigned long long job_token = 0;
    std::string rendezvous;
    iile::IisptOptions iispt;
    int integrator_choice = -1;   // --integrator: IILE_INTEGRATOR_*; -1: the scene file's
    if (const char *e = getenv("IILE_IISPT_NET")) iispt.net_file = e;
    for (int i = 1; i < argc; ++i) {
        auto arg_int = [&](int &dst) {
            if (i + 1 < argc) dst = atoi(argv[++i]);
        };
        // pbrt's own options keep their spellings (src/main/pbrt.cpp:106-186): --outfile, --quick, --quiet, --nthreads and
        // the logging flags are accepted (the last three change nothing here: no thread pool, no glog)
        if ((!strcmp(argv[i], "--outfile") || !strcmp(argv[i], "-outfile")) && i + 1 < argc)
            out = argv[++i];
        else if (!strncmp(argv[i], "--outfile=", 10))
            out = argv[i] + 10;
        else if (!strcmp(argv[i], "--quick") || !strcmp(argv[i], "-quick"))
            ps.quick = true;
        else if (!strcmp(argv[i], "--quiet") || !strcmp(argv[i], "-quiet"))
            quiet = true;
        else if (!strcmp(argv[i], "--logtostderr") || !strncmp(argv[i], "--nthreads=", 11) || !strncmp(argv[i], "--logdir=", 9) ||
                 !strncmp(argv[i], "--minloglevel=", 14) || !strncmp(argv[i], "--v=", 4))
            ;
        else if ((!strcmp(argv[i], "--nthreads") || !strcmp(argv[i], "-nthreads") || !strcmp(argv[i], "--logdir") || !strcmp(argv[i], "-logdir") ||
                  !strcmp(argv[i], "--minloglevel") || !strcmp(argv[i], "-minloglevel") || !strcmp(argv[i], "--v") || !strcmp(argv[i], "-v")) &&
                 i + 1 < argc)
            ++i;
        else if (!strncmp(argv[i], "--iileIndirect=", 15))
            iispt.indirect_tasks = atoi(argv[i] + 15);
        else if (!strncmp(argv[i], "--iileDirect=", 13))
            iispt.direct_samples = atoi(argv[i] + 13);
        else if (!strncmp(argv[i], "--iispt_hemi_size=", 18))
            iispt.hemi_size = atoi(argv[i] + 18);
        else if (!strncmp(argv[i], "--iisptNet=", 11))
            iispt.net_file = argv[i] + 11;
        else if (!strncmp(argv[i], "--iisptIndirectOut=", 19))
            iispt.indirect_out = argv[i] + 19;
        else if (!strncmp(argv[i], "--iisptDirectOut=", 17))
            iispt.direct_out = argv[i] + 17;
        else if (!strcmp(argv[i], "--integrator") && i + 1 < argc) {
            const char *in = argv[++i];
            if (strcmp(in, "path") && strcmp(in, "iispt")) {
                fprintf(stderr, "iile_pbrt: --integrator wants path or iispt\n");
                return 1;
            }
            integrator_choice = !strcmp(in, "iispt") ? IILE_INTEGRATOR_IISPT : IILE_INTEGRATOR_PATH;
        }
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
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) {
            gpus = atoi(argv[++i]);
            gpus_given = true;
            if (gpus < 1) {
                fprintf(stderr, "iile_pbrt: --gpus wants a number >= 1\n");
                return 1;
            }
        }
        else if (!strcmp(argv[i], "--sampler") && i + 1 < argc) {
            const char *sn = argv[++i];
            ps.sampler = !strcmp(sn, "sobol") ? IILE_SAMPLER_SOBOL : (!strcmp(sn, "halton") ? IILE_SAMPLER_HALTON : IILE_SAMPLER_KEEP);
        } else if (!strcmp(argv[i], "--splitmethod") && i + 1 < argc) {
            const char *sm = argv[++i];
            ps.splitmethod = !strcmp(sm, "hlbvh") ? IILE_SPLIT_HLBVH : (!strcmp(sm, "middle") ? IILE_SPLIT_MIDDLE : (!strcmp(sm, "equal") ? IILE_SPLIT_EQUAL : IILE_SPLIT_SAH));
        } else if (!strcmp(argv[i], "--bvh-device"))
            ps.bvh_on_device = true;
        else if (!strcmp(argv[i], "--gpurank") && i + 1 < argc) {
            if (sscanf(argv[++i], "%d/%d", &gpu_rank, &gpu_nranks) != 2 || gpu_nranks < 1 || gpu_rank < 0 || gpu_rank >= gpu_nranks) {
                fprintf(stderr, "iile_pbrt: --gpurank wants R/N with 0 <= R < N\n");
                return 1;
            }
            ranked = true;
        } else if (!strcmp(argv[i], "--rendezvous") && i + 1 < argc)
            rendezvous = argv[++i];
        else if (!strcmp(argv[i], "--job") && i + 1 < argc) {
            // decimal only, the whole argument, not 0 (0 means "no token"): a token that starts with 0 read in base 0 is octal
            // and stops at an 8 or 9, and garbage would silently become the token-less mode
            const char *arg = argv[++i];
            char *end = nullptr;
            errno = 0;
            job_token = strtoull(arg, &end, 10);
            if (errno != 0 || end == arg || *end != '\0' || job_token == 0 || arg[0] == '-') {
                fprintf(stderr, "iile_pbrt: --job wants a decimal number > 0 (the same for every rank of one launch), got \"%s\"\n", arg);
                return 1;
            }
        }
        else if (argv[i][0] == '-') {
            fprintf(stderr, "usage: iile_pbrt scene.pbrt [--outfile f.exr|f.pfm] [--quick] [--quiet] [--nthreads N] [--xres N] [--yres N] [--spp N] "
                            "[--maxdepth N] [--stats] [--gpus N] [--sampler halton|sobol] [--splitmethod sah|hlbvh|middle|equal] [--bvh-device] "
                            "[--gpurank R/N --rendezvous FILE [--job TOKEN]] [--integrator path|iispt] [--iisptNet=FILE] [--iileIndirect=TASKS] "
                            "[--iileDirect=SAMPLES] [--iispt_hemi_size=32] [--iisptIndirectOut=FILE] [--iisptDirectOut=FILE]\n");
            return 1;
        } else
            scene_file = argv[i];
    }
    // the fork's sampler override: `path` renders with a SobolSampler of that many samples (src/integrators/path.cpp:202-212)
    if (const char *ovr = getenv("IILE_PATH_SAMPLES_OVERRIDE")) {
        ps.sampler = IILE_SAMPLER_SOBOL;
        ps.pixelsamples = atoi(ovr);
        fprintf(stderr, "iile_pbrt: Created override sampler with [%d] spp\n", ps.pixelsamples);
    }
    if (scene_file.empty()) {
        fprintf(stderr, "iile_pbrt: no scene file given\n");
        return 1;
    }
    iile_dist *comm = nullptr;
    if (ranked) {
        if (rendezvous.empty()) {
            fprintf(stderr, "iile_pbrt: --gpurank R/N needs --rendezvous FILE\n");
            return 1;
        }
        const int n_dev = iile_device_count();
        if (n_dev < 1 || iile_device_select(gpu_rank % n_dev) != IILE_OK) {
            fprintf(stderr, "Error: GPU path: %s\n", n_dev < 1 ? "no HIP device" : iile_last_error());
            return 1;
        }
        // no launcher tears this job down if a rank is missing: the communicator carries a deadline ($IILE_DIST_TIMEOUT_S, default
        // 120 s) — set-up and every later wait on it end with an error instead of hanging (include/iile_dist.h)
        double timeout_s = 120.0;
        if (const char *e = getenv("IILE_DIST_TIMEOUT_S")) timeout_s = atof(e) > 0 ? atof(e) : timeout_s;
        uint8_t id[IILE_DIST_ID_BYTES];
        if (iile_dist_rendezvous_file_token(rendezvous.c_str(), gpu_rank, job_token, id, int(timeout_s + 0.999)) != IILE_OK ||
            iile_dist_create_deadline(id, gpu_rank, gpu_nranks, timeout_s, &comm) != IILE_OK ||
            iile_dist_rendezvous_done(comm, rendezvous.c_str()) != IILE_OK) {
            fprintf(stderr, "Error: multi-GPU set-up: %s\n", iile_dist_last_error());
            if (comm) iile_dist_abort(comm);
            return 1;
        }
    }
    lap("arguments read");
    iile::Scene scene(scene_file, ps);
    lap("scene file parsed, BVH built (host)");
    if (out.empty()) out = scene.ok() ? scene.film_filename() : std::string("pbrt.exr");
    if (integrator_choice < 0) integrator_choice = scene.ok() ? scene.integrator() : IILE_INTEGRATOR_PATH;
    if (integrator_choice == IILE_INTEGRATOR_IISPT) {
        if (ranked && gpus_given) {
            fprintf(stderr, "iile_pbrt: --gpurank (one process per GPU) and --gpus (one process, several devices) exclude each other\n");
            if (comm) iile_dist_destroy(comm);
            return 1;
        }
        if (ranked) iispt.rank = gpu_rank, iispt.nranks = gpu_nranks, iispt.comm = comm;
        if (const char *e = getenv("IILE_DEBUG_IISPT_SHARD")) {   // tests: the share of rank R of N alone, no communicator ("R/N")
            int r = 0, n = 1;
            if (!ranked && sscanf(e, "%d/%d", &r, &n) == 2 && n >= 1 && r >= 0 && r < n) iispt.rank = r, iispt.nranks = n;
        }
        std::unique_ptr<iile::GpuIisptIntegrator> ii(iile::CreateGpuIisptIntegrator(ps, out, iispt));
        // --gpus N: one process, N devices (a host thread each; --gpus 1 is the same code with a communicator of one rank)
        const bool ok_ii = (gpus_given && !ranked) ? ii->RenderAllDevices(scene, gpus) : ii->Render(scene);
        if (comm) {
            if (ok_ii) iile_dist_destroy(comm);
            else iile_dist_abort(comm);
        }
        if (!ok_ii) return 1;
        lap("IISPT frame rendered and written");
        if (gpu_rank != 0) return 0;
        if (!quiet)
            printf("IISPT: %d tasks, %lld hemi points, %lld probes, %lld pixels gathered, %d direct passes -> %s\n", ii->stats.tasks, ii->stats.hemi_points,
                   ii->stats.probes, ii->stats.pixels, iispt.direct_samples, out.c_str());
        return 0;
    }
    std::unique_ptr<iile::GpuPathIntegrator> integrator(iile::CreateGpuPathIntegrator(ps, out, 0, 1, stats, comm));
    if (!ranked) integrator->UseDevices(gpus_given ? gpus : 0);
    const bool ok = integrator->Render(scene);
    lap("frame rendered and written");
    if (comm) {
        if (ok) iile_dist_destroy(comm);
        else iile_dist_abort(comm);   // (the job is over: do not wait for the other ranks in ncclCommDestroy)
    }
    if (!ok) return 1;
    if (gpu_rank != 0) return 0;
    const iile_stats &st = integrator->last_stats;
    if (!quiet) printf("rendered %llu camera samples in %.1f ms -> %s\n", (unsigned long long)st.n_paths, st.ms_total, out.c_str());
    if (stats)
        printf("rays: %llu closest + %llu shadow; BVH nodes visited %llu + %llu; triangle tests %llu (%llu hits)\n",
               (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays,
               (unsigned long long)st.nodes_closest, (unsigned long long)st.nodes_any,
               (unsigned long long)st.tri_tests, (unsigned long long)st.tri_hits);
    return 0;
}
