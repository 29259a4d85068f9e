/*
 * iile_scene.h — the flattened, POD scene description that crosses the C ABI.
 *
 * This is the data the reference keeps behind `Scene`, `BVHAccel` (private
 * arrays, /root/reference/src/accelerators/bvh.h:90-94), `TriangleMesh`,
 * `Sphere`, `Material`, `DiffuseAreaLight`, `PerspectiveCamera`, `Film` and
 * `HaltonSampler` objects, laid out as plain arrays so that a GPU library (or
 * the CPU oracle used by the tests) can consume it without any C++ types.
 *
 * Producer : libiile_host.so  (iile_host.h: iile_host_load_pbrt)
 * Consumers: libiile_gpu.so   (iile_gpu.h:  iile_scene_create)
 *            oracle/          (tests only)
 *
 * All matrices are row-major float[16] (m[r][c] = a[4*r+c]), exactly the
 * reference's Matrix4x4 (src/core/transform.h:60-110).
 */
#ifndef IILE_SCENE_H
#define IILE_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the few inline helpers of this header are also called from the HIP kernels */
#if defined(__HIPCC__)
#define IILE_INLINE static inline __host__ __device__
#else
#define IILE_INLINE static inline
#endif

/* Mirrors LinearBVHNode, src/accelerators/bvh.cpp:95-104 (32 bytes). */
typedef struct iile_bvh_node {
    float bmin[3];
    float bmax[3];
    int32_t offset;  /* leaf: first primitive; interior: second child index */
    uint16_t nprims; /* 0 -> interior */
    uint8_t axis;    /* interior: split axis */
    uint8_t pad;
} iile_bvh_node;

/* prim_flags bits (one u32 per primitive, primitives are in BVH leaf order,
 * i.e. the order of BVHAccel::primitives after the build). */
enum {
    IILE_PRIM_SPHERE = 1u << 0,      /* else triangle */
    IILE_PRIM_HAS_NORMALS = 1u << 1, /* TriangleMesh::n != null */
    IILE_PRIM_HAS_UV = 1u << 2,      /* TriangleMesh::uv != null */
    IILE_PRIM_FLIP = 1u << 3,        /* reverseOrientation ^ transformSwapsHandedness */
    IILE_PRIM_HAS_ALPHA = 1u << 4    /* TriangleMesh::alphaMask or shadowAlphaMask: see prim_alpha */
};
/* prim_alpha values: an image texture index (a "float" imagemap: all three channels of its texels hold
 * the float), or */
#define IILE_ALPHA_NONE (-1) /* no mask (or a constant non-zero one: it never rejects) */
#define IILE_ALPHA_ZERO (-2) /* ConstantTexture<Float>(0): every hit is rejected */

/* Sphere, src/shapes/sphere.h:47-76 + Shape base (src/core/shape.cpp:45-52). */
typedef struct iile_sphere {
    float o2w[16], o2w_inv[16]; /* ObjectToWorld.m / .mInv; WorldToObject is the swap */
    float radius, zmin, zmax, theta_min, theta_max, phi_max;
    int32_t reverse_orientation;
    int32_t swaps_handedness;
} iile_sphere;

enum { IILE_MAT_MATTE = 0, IILE_MAT_PLASTIC = 1, IILE_MAT_UBER = 2, IILE_MAT_MIRROR = 3, IILE_MAT_GLASS = 4 };

/* MatteMaterial / PlasticMaterial / UberMaterial / MirrorMaterial / GlassMaterial with constant
 * textures (src/materials/matte.cpp:45-62, plastic.cpp:45-70, uber.cpp:45-100, mirror.cpp:44-55,
 * glass.cpp:45-92). Uber: its two SpecularTransmission lobes (the pass-through of opacity < 1 and Kt,
 * uber.cpp:53-61, 94-99) are rendered by every entry point (the path and probe passes, the IISPT runner's stages, the direct pass).
 * Glass: smooth (uroughness = vroughness = 0: one FresnelSpecular lobe, as the path integrator gets it) or rough
 * (uroughness or vroughness != 0: MicrofacetReflection + MicrofacetTransmission, glass.cpp:66-90). */
typedef struct iile_material {
    int32_t type;
    float kd[3];     /* matte, plastic, uber; 0 for mirror */
    float ks[3];     /* plastic, uber: glossy (microfacet) reflectance */
    float sigma;     /* matte: Clamp(sigma, 0, 90) degrees; 0 = Lambertian, else Oren-Nayar (reflection.h:410-427) */
    float roughness; /* plastic, uber: as given (uber: "uroughness" if given); glass: uroughness */
    float alpha;     /* plastic, uber: RoughnessToAlpha(roughness) if remap else roughness
                        (src/core/microfacet.h:123-128) */
    int32_t remap_roughness;
    float eta;       /* uber, glass: index of refraction of FresnelDielectric(1, eta) */
    float kr[3];     /* uber, mirror, glass: specular reflectance */
    float kt[3];     /* glass, uber: specular transmittance */
    float on_a, on_b; /* matte with sigma != 0: the Oren-Nayar constants A, B (reflection.h:416-419) */
    /* image textures (index into iile_scene_desc::textures) whose value at a hit, multiplied by the constant kd /
     * ks / kr / kt (1 for a plain "imagemap", the constant factor of a "scale" texture, textures/scale.h:56-58),
     * replaces that constant, or -1: Texture<Spectrum>::Evaluate of an ImageTexture (textures/imagemap.h:87-94) */
    int32_t kd_tex, ks_tex, kr_tex, kt_tex;
    /* "bumpmap": a float image texture (its float in all three channels of the texels) displacing the shading
     * geometry at a hit (Material::Bump, src/core/material.cpp:45-86), or -1 */
    int32_t bump_tex;
    /* "roughness" of plastic / uber as a float image texture, looked up at the hit and mapped by RoughnessToAlpha
     * if remap_roughness (plastic.cpp:60-63, uber.cpp:79-84, microfacet.h:123-128), or -1 */
    int32_t rough_tex;
    /* "sigma" of matte as a float image texture (degrees, clamped to [0, 90] at the hit: matte.cpp:56-61), or -1 */
    int32_t sigma_tex;
    /* uber: "opacity" (constant; {1, 1, 1} for every other material): 1 - opacity is the pass-through lobe, every other
     * coefficient is multiplied by it (uber.cpp:53-99) */
    float opacity[3];
    /* uber, glass: "vroughness" and its alpha where it differs from "uroughness" (roughness / alpha above): an anisotropic
     * TrowbridgeReitzDistribution(alphax, alphay) (uber.cpp:73-86, glass.cpp:52-73); equal to roughness / alpha otherwise */
    float roughness_v, alpha_v;
    /* uber: "opacity" as an image texture (times the constant `opacity`, as kd_tex .. kt_tex), or -1 */
    int32_t opacity_tex;
    /* uber: where alpha_v comes from at a hit — -1: the constant alpha_v above ("vroughness" given as a number); -2: whatever the hit's
     * alpha along u is ("vroughness" not given: roughv = roughu, uber.cpp:83-84), the value for plastic too; >= 0: a float image texture
     * for "vroughness", mapped like rough_tex. rough_tex is then "uroughness" if that parameter is given, else "roughness" (uber.cpp:79-82) */
    int32_t rough_tex_v;
} iile_material;

/* ImageTexture<RGBSpectrum, Spectrum> over a UVMapping2D (src/textures/imagemap.h:78-112,
 * src/core/texture.cpp:170-180) with its MIPMap (src/core/mipmap.h:61-110): the pyramid is built on the
 * host exactly as MIPMap's constructor does (Lanczos resampling to powers of two, 2x2 box levels) and
 * handed over as plain arrays. Level l is level_w[l] x level_h[l] RGB texels, row-major (t * w + s), row 0 =
 * the image's BOTTOM scanline (imagemap.cpp:67-74 flips), starting at float index 3 * level_offset[l]
 * of iile_scene_desc::texels. */
#define IILE_MAX_TEX_LEVELS 16
#define IILE_WRAP_REPEAT 0
#define IILE_WRAP_BLACK 1
#define IILE_WRAP_CLAMP 2
#define IILE_EWA_LUT_SIZE 128
typedef struct iile_texture {
    int32_t n_levels;
    int32_t wrap;      /* IILE_WRAP_* */
    int32_t trilinear; /* doTrilinear */
    float max_aniso;
    float su, sv, du, dv; /* UVMapping2D */
    int32_t level_w[IILE_MAX_TEX_LEVELS], level_h[IILE_MAX_TEX_LEVELS];
    int64_t level_offset[IILE_MAX_TEX_LEVELS]; /* in texels */
} iile_texture;

/* A light: DiffuseAreaLight on a shape (src/lights/diffuse.h:48-75) or one of the delta lights
 * PointLight (src/lights/point.h:49-70), SpotLight (src/lights/spot.h:49-74), DistantLight
 * (src/lights/distant.h:49-72). */
#define IILE_MAX_LIGHTS 8 /* the device keeps per-voxel light distributions for this many */
#define IILE_LIGHT_DIFFUSE_AREA 0
#define IILE_LIGHT_POINT 1
#define IILE_LIGHT_SPOT 2
#define IILE_LIGHT_DISTANT 3
#define IILE_LIGHT_AREA_TRIANGLE 4 /* DiffuseAreaLight on one triangle (every triangle of an emitting mesh is a light) */
#define IILE_LIGHT_INFINITE 5      /* InfiniteAreaLight without an environment map (src/lights/infinite.h:51-84) */
typedef struct iile_light {
    float lemit[3];  /* area: Lemit (L * scale); point, spot: I * scale; distant: L * scale */
    int32_t two_sided;
    int32_t sphere;  /* area light on a sphere: index into spheres[]; else -1 */
    int32_t type;    /* IILE_LIGHT_* */
    float pos[3];    /* point, spot: pLight = LightToWorld(0,0,0); distant: wLight = Normalize(LightToWorld(dir)) */
    float w2l[9];    /* spot: rows of the upper 3x3 of WorldToLight (for Falloff, spot.cpp:66-76) */
    float cos_total_width, cos_falloff_start; /* spot */
    float world_radius; /* distant: radius of the scene's bounding sphere (Light::Preprocess, distant.cpp:63-65) */
    int32_t prim;       /* triangle area light: its primitive, in BVH order */
    /* Light::nSamples (area and infinite lights: "samples" / "nsamples", diffuse.cpp:140-141, infinite.cpp:181-182; 0 means 1).
     * The path integrator never looks at it (UniformSampleOneLight); the IISPT direct pass's UniformSampleAllLights does
     * (integrator.cpp:54-83): iile_render_direct takes n_samples light samples per vertex from every light, at most 64 per
     * vertex over all lights (it rejects a scene that asks for more). pbrt --quick divides it by 4 (diffuse.cpp:143,
     * infinite.cpp:183: iile_host_overrides::quick_render). */
    int32_t n_samples;
    /* infinite: lemit is L * scale; w2l as for the spot light; world_radius as for the distant light; and */
    float l2w[9];       /* upper 3x3 of LightToWorld, row major */
    /* Lmap (infinite.cpp:49-63): index into iile_scene_desc::textures of the pyramid of the environment map
     * times L (one texel without a map; not flipped in y; repeat wrap) */
    int32_t env_tex;
    /* the Distribution2D over the dist_w x dist_h (= 2 * the map's size) image of filtered luminance *
     * sin(theta) (infinite.cpp:65-83, sampling.cpp:159-174), at float index dist_offset of
     * iile_scene_desc::env_dist: per row v the conditional Distribution1D {func[w], cdf[w + 1], funcInt},
     * then the marginal one {func[h], cdf[h + 1], funcInt} over the rows' funcInt */
    int32_t dist_w, dist_h;
    int64_t dist_offset;
} iile_light;

/* PerspectiveCamera (src/cameras/perspective.cpp:50-72, src/core/camera.h:90-111). */
typedef struct iile_camera {
    float raster_to_camera[16];
    float camera_to_world[16];
    float lens_radius, focal_distance, shutter_open, shutter_close;
    /* dxCamera, dyCamera: RasterToCamera((1,0,0)) - RasterToCamera((0,0,0)) and the same for y
     * (perspective.cpp:58-62); the ray differentials of camera rays are built from them */
    float dx_camera[3], dy_camera[3];
} iile_camera;

/* Film + box filter (src/core/film.cpp:45-91, src/filters/box.cpp:41-47). */
typedef struct iile_film_desc {
    int32_t xres, yres;              /* fullResolution */
    int32_t crop_x0, crop_y0, crop_x1, crop_y1; /* croppedPixelBounds */
    int32_t samp_x0, samp_y0, samp_x1, samp_y1; /* GetSampleBounds() */
    float filter_rx, filter_ry;      /* box filter radius */
    float scale;
    float max_sample_luminance;
} iile_film_desc;

/* HaltonSampler state (src/samplers/halton.cpp:65-127). */
typedef struct iile_halton {
    int32_t spp;
    int32_t base_scales[2];
    int32_t base_exponents[2];
    int32_t sample_stride;
    int32_t mult_inverse[2];
    int32_t n_dims;            /* number of prime bases covered by perms */
    const uint16_t *perms;     /* concatenated digit permutations, PrimeSums layout */
    const int32_t *primes;     /* [n_dims] */
    const int32_t *prime_sums; /* [n_dims] */
    int32_t n_perms;           /* total u16 entries */
    int32_t sample_at_pixel_center; /* "samplepixelcenter": dimensions 0 and 1 are 0.5 (halton.cpp:119) */
} iile_halton;

/* SobolSampler state (src/samplers/sobol.h:57-75, sobol.cpp:42-59; the sampler the fork's CreatePathIntegrator puts in
 * place of the scene's under IILE_PATH_SAMPLES_OVERRIDE, src/integrators/path.cpp:202-212). enabled != 0: the frame is
 * sampled with it and iile_halton is unused (the IISPT probe pass keeps its own Halton sampler). Sample indices are
 * confined to 32 bits (spp << 2 log2_resolution <= 2^32), so 32 of the 52 columns of each generator matrix suffice. */
typedef struct iile_sobol {
    int32_t enabled;
    int32_t spp;                /* RoundUpPow2(pixelsamples) */
    int32_t resolution;         /* RoundUpPow2(max extent of the sample bounds) */
    int32_t log2_resolution;
    int32_t n_dims;
    const uint32_t *matrices32; /* [n_dims * 32]: columns 0 .. 31 of SobolMatrices32 (src/core/sobolmatrices.cpp) per dimension */
    uint32_t vdc[32];           /* VdCSobolMatrices[log2_resolution - 1][c] */
    uint32_t vdc_inv[32];       /* VdCSobolMatricesInv[log2_resolution - 1][c] */
} iile_sobol;

/* PathIntegrator knobs (src/integrators/path.cpp:214-231). */
#define IILE_LIGHTS_SPATIAL 0 /* SpatialLightDistribution, the default (lightdistrib.cpp:91-299) */
#define IILE_LIGHTS_UNIFORM 1 /* UniformLightDistribution (lightdistrib.cpp:65-72) */
#define IILE_LIGHTS_POWER 2   /* PowerLightDistribution over Light::Power().y() (integrator.cpp:217-225) */
typedef struct iile_integrator {
    int32_t max_depth;
    float rr_threshold;
    int32_t light_strategy;             /* "lightsamplestrategy" of the path integrator (path.cpp:231), IILE_LIGHTS_* */
    float light_power[IILE_MAX_LIGHTS]; /* power strategy: Power().y() of every light */
    /* "pixelbounds" (path.cpp:216-229): {x0, y0, x1, y1} = Intersect(film->GetSampleBounds(), the four values given); pixels of the
     * sample bounds outside it take no samples (SamplerIntegrator::Render, integrator.cpp:272). The sample bounds when not given;
     * all four zero = not given (what a caller that does not know the field leaves); the host writes an empty intersection as {0, 0, -1, -1} */
    int32_t pixel_bounds[4];
} iile_integrator;

/* The IISPT probe pass: what IISPTdIntegrator::RenderView renders from a HemisphericCamera
 * (src/integrators/iispt_d.cpp:66-470, src/cameras/hemispheric.cpp:15-160): a hemi_size x hemi_size film behind a
 * Gaussian filter (radius 2, alpha 2; CreateHemisphericCamera), one Halton sample per pixel from a sampler built
 * for the film's sample bounds (CreateIISPTdIntegrator with iileDSampler "halton"), maxdepth 3. The host fills
 * this once; the probes' cameras (position, direction) arrive per call. */
typedef struct iile_probe_setup {
    int32_t hemi_size;       /* PbrtOptions.iisptHemiSize, 32 */
    int32_t max_depth;       /* 3, hard-coded in CreateIISPTdIntegrator */
    iile_film_desc film;     /* hemi_size^2, crop = full, sample bounds -2 .. hemi_size + 2 */
    float filter_table[256]; /* GaussianFilter((2, 2), 2) */
    int32_t base_scales[2], base_exponents[2], sample_stride, mult_inverse[2]; /* HaltonSampler(1, sampleBounds) */
} iile_probe_setup;

typedef struct iile_scene_desc {
    /* BVH, depth-first flattened (src/accelerators/bvh.cpp:640-658) */
    int32_t n_nodes;
    const iile_bvh_node *nodes;
    /* primitives in BVH order */
    int32_t n_prims;
    const uint32_t *prim_flags;    /* [n_prims] */
    const int32_t *prim_material;  /* [n_prims] index into materials */
    const int32_t *prim_light;     /* [n_prims] index into lights or -1 */
    const int32_t *prim_shape;     /* [n_prims] sphere index for spheres, mesh id for triangles */
    const float *tri_p;            /* [n_prims*9] world-space p0,p1,p2 (unused for spheres) */
    const float *tri_n;            /* [n_prims*9] world-space vertex normals (if HAS_NORMALS) */
    const float *tri_uv;           /* [n_prims*6] (if HAS_UV) */
    const int32_t *prim_alpha;     /* [n_prims*2] {alphaMask, shadowAlphaMask} of the primitive's mesh
                                      (triangle.cpp:325-331, 509-541); NULL when no primitive has HAS_ALPHA */
    int32_t n_spheres;
    const iile_sphere *spheres;
    int32_t n_materials;
    const iile_material *materials;
    int32_t n_lights;
    const iile_light *lights;
    int64_t n_env_dist;            /* floats in env_dist */
    const float *env_dist;         /* sampling distributions of the infinite lights (iile_light::dist_offset) */
    int32_t n_textures;
    const iile_texture *textures;
    int64_t n_texels;              /* RGB texels in all levels of all textures */
    const float *texels;           /* [3 * n_texels] */
    float ewa_lut[IILE_EWA_LUT_SIZE]; /* MIPMap::weightLut, mipmap.h:199-205 */
    /* Film::filterTable (film.cpp:65-74): filter->Evaluate at the centres of a 16 x 16 grid over the positive
     * quadrant of its support, for box / gaussian / mitchell / sinc / triangle (src/filters). film_filter_wide = 0
     * for the box filter of radius 0.5 (every sample lands in its own pixel; the fast film kernels), 1 otherwise */
    int32_t film_filter_wide;
    float film_filter_table[256];
    iile_camera camera;
    iile_film_desc film;
    iile_halton halton;
    iile_integrator integrator;
    iile_probe_setup probe;
    iile_sobol sobol;
} iile_scene_desc;

/* One task of the IISPT render runner (IisptScheduleMonitorTask, src/integrators/iisptschedulemonitor.cpp:40-79): the
 * film pixels [x0, x1) x [y0, y1) with hemi points every `tilesize` pixels, at x0, x0 + tilesize, ... and at x1 - 1
 * (likewise in y), visited row by row (src/integrators/iisptrenderrunner.cpp:248-260, 387-412). */
typedef struct iile_iispt_task {
    int32_t x0, y0, x1, y1;
    int32_t tilesize;
    uint32_t counter_base; /* IisptRenderRunner::sampler_pixel_counter.x before the task (iisptrenderrunner.cpp:941-953):
                              the task's i-th camera sample (hemi points first, then film pixels row-major) is taken with
                              the sampler on pixel (counter_base + 1 + i, 0) */
    uint64_t rng_seed;     /* film pixel j of the task (row-major) draws from pbrt's RNG(rng_seed + j); the reference uses one
                              RNG per thread, consumed in scheduling order — not reproducible, nor needed, in parallel */
} iile_iispt_task;
/* number of hemi points of a task along one axis of extent [a0, a1) */
IILE_INLINE int32_t iile_iispt_grid_count(int32_t a0, int32_t a1, int32_t tilesize) {
    int32_t n = 1, t = a0;
    if (a1 <= a0 || tilesize < 1) return 0;
    while (t != a1 - 1) {
        t = t + tilesize < a1 - 1 ? t + tilesize : a1 - 1;
        ++n;
    }
    return n;
}
/* position of hemi point i along that axis, and the index of the hemi point at position t */
IILE_INLINE int32_t iile_iispt_grid_pos(int32_t a0, int32_t a1, int32_t tilesize, int32_t i) {
    const int32_t t = a0 + i * tilesize;
    return t < a1 - 1 ? t : a1 - 1;
}
IILE_INLINE int32_t iile_iispt_grid_index(int32_t a0, int32_t a1, int32_t tilesize, int32_t t) {
    return t == a1 - 1 ? iile_iispt_grid_count(a0, a1, tilesize) - 1 : (t - a0) / tilesize;
}

/* Which rank of an n-rank job renders the 16x16 tile (tx, ty) of SamplerIntegrator::Render's tile grid
 * (src/core/integrator.cpp:235-248: tiles are independent units of work). Diagonal interleave: every run of n
 * consecutive tiles of a tile row OR column holds one tile of each rank, for any n (an interleave of the linear
 * tile index would collapse to vertical stripes whenever the tiles per row are a multiple of n: 1080p has 120).
 * The map is part of the boundary: libiile_gpu.so, the C++ host and the test oracle all use this one definition. */
IILE_INLINE int32_t iile_tile_owner(int32_t tx, int32_t ty, int32_t nranks) {
    return nranks <= 1 ? 0 : (int32_t)(((uint32_t)tx + (uint32_t)ty) % (uint32_t)nranks);
}

#ifdef __cplusplus
}
#endif
#endif /* IILE_SCENE_H */
