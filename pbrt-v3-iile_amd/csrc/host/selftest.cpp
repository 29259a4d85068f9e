// selftest.cpp — the host library under AddressSanitizer / UBSan (the reference's sanitizer builds have no
// counterpart on the GPU, so the CPU side carries one). `make asan` links this driver with the host sources compiled
// -fsanitize=address,undefined; it loads every scene file given on the command line through the C ABI (parser, Loop
// subdivision, PLY / image readers, MIP pyramids, BVH build, sampler tables incl. the Sobol' matrices), touches the
// flattened arrays, converts a film and frees everything. Files that are expected to be rejected are prefixed `!`; image
// files to push through ReadImage / WriteImageEXR (well-formed or not) are prefixed `@`.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/iile_host.h"

int main(int argc, char **argv) {
    int failures = 0;
    for (int i = 1; i < argc; ++i) {
        if (argv[i][0] == '@') {  // an image file through ReadImage (it may be malformed: any outcome but a sanitizer report)
            int32_t w = 0, h = 0;
            int rc = iile_host_read_image(argv[i] + 1, &w, &h, nullptr);
            if (rc == 0) {
                std::vector<float> px(3 * size_t(w) * size_t(h));
                rc = iile_host_read_image(argv[i] + 1, &w, &h, px.data());
                if (rc == 0 && !px.empty()) {  // and back out through WriteImageEXR
                    const std::string out = std::string(argv[i] + 1) + ".copy.exr";
                    if (iile_host_write_exr(out.c_str(), px.data(), 1, 2, 1 + w, 2 + h, w + 5, h + 5) != 0) ++failures;
                }
            }
            printf("selftest: image %s: %s (%d x %d)\n", argv[i] + 1, rc == 0 ? "read" : iile_host_last_error(), w, h);
            continue;
        }
        const bool expect_error = argv[i][0] == '!';
        const char *path = argv[i] + (expect_error ? 1 : 0);
        for (int sampler = 1; sampler <= 2; ++sampler) {
            iile_host_overrides ov = {64, 48, 4, 0, sampler};
            iile_host_scene *hs = nullptr;
            const int rc = iile_host_load_pbrt(path, &ov, &hs);
            if (rc != 0) {
                if (!expect_error) {
                    fprintf(stderr, "selftest: %s failed to load: %s\n", path, iile_host_last_error());
                    ++failures;
                }
                break;
            }
            if (expect_error) {
                fprintf(stderr, "selftest: %s loaded although it should have been rejected\n", path);
                ++failures;
            }
            const iile_scene_desc *d = iile_host_scene_desc(hs);
            iile_host_scene_info info;
            iile_host_scene_get_info(hs, &info);
            // walk the arrays the consumers read
            double sum = 0;
            for (int n = 0; n < d->n_nodes; ++n) sum += d->nodes[n].bmin[0] + d->nodes[n].bmax[2];
            for (int p = 0; p < d->n_prims; ++p) sum += d->tri_p[9 * size_t(p)] + d->tri_n[9 * size_t(p) + 8] + d->tri_uv[6 * size_t(p) + 5] + d->prim_flags[p];
            for (int64_t t = 0; t < d->n_texels; ++t) sum += d->texels[3 * size_t(t)];
            for (int q = 0; q < d->halton.n_perms; ++q) sum += d->halton.perms[q];
            if (d->sobol.enabled)
                for (int q = 0; q < d->sobol.n_dims * 32; ++q) sum += d->sobol.matrices32[q];
            const iile_film_desc *f = iile_host_scene_film(hs);
            const size_t n_pix = size_t(f->crop_x1 - f->crop_x0) * size_t(f->crop_y1 - f->crop_y0);
            std::vector<float> xyzw(4 * n_pix, 1.f), rgb(3 * n_pix);
            iile_host_film_to_rgb(f, xyzw.data(), rgb.data());
            printf("selftest: %s sampler %d: %d prims, %d nodes, %d textures (checksum %.6g)\n", path, sampler, info.n_prims, info.n_nodes,
                   d->n_textures, sum + rgb[0]);
            iile_host_scene_free(hs);
        }
    }
    std::vector<uint32_t> m32(1024 * 52);
    std::vector<uint64_t> m64(1024 * 52), vdc(52), inv(52);
    if (iile_host_sobol_matrices(1024, m32.data(), m64.data()) != 0 || iile_host_sobol_vdc(11, vdc.data(), inv.data()) != 0) ++failures;
    printf("selftest: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
