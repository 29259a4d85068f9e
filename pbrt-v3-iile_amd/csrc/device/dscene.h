// dscene.h — device-resident scene layout (HBM) and queue records.
//
// Layout choices (DESIGN.md "Data layout in HBM"):
//  * The BVH keeps the reference's tree (bvh.cpp:640-658) but interior nodes are
//    re-packed into 64-byte records holding both children's boxes; leaves have no
//    record (a leaf reference is ~firstPrimitive, the last primitive of a leaf
//    is flagged in its vertex record).
//  * Triangle vertices are pre-gathered per primitive in BVH leaf order:
//    3 x float4 = 48 contiguous bytes per triangle test, with the primitive's
//    flags / material / light packed into the .w lanes, replacing the
//    reference's shared_ptr<Primitive> -> Triangle -> mesh->p[v[i]] chase.
//  * Ray / hit / NEE queues are arrays of float4 records so that a wavefront's
//    64 lanes issue full-width 16-byte coalesced loads and stores.
#pragma once
#include "dmath.h"

namespace iile {

struct DSphere {
    M44 o2w, o2w_inv;
    float radius, zmin, zmax, theta_min, theta_max, phi_max;
    int reverse_orientation, swaps_handedness;
    // (*ObjectToWorld)(Point3f(0, 0, 0)), sphere.cpp:254 / :296, worked out once at upload with xf_point's own operations
    // (api.hip): the light-sampling code of every hit starts from it
    float center[3];
    float pad_;
};
enum { kMatMatte = 0, kMatPlastic = 1, kMatUber = 2, kMatMirror = 3, kMatGlass = 4 };  // = IILE_MAT_* (checked in api.hip)
// (16-byte aligned, and the fields every hit reads first: (type, kd) and (ks, alpha) are one float4 each, see make_bsdf)
struct alignas(16) DMaterial {
    int type;
    float kd[3];
    float ks[3];
    float alpha;  // plastic, uber: the microfacet distribution's alpha; glass: the same, 0 = smooth
    float kr[3];  // uber, mirror, glass: specular reflectance
    float eta;    // uber, glass: FresnelDielectric(1, eta)
    float kt[3];  // glass, uber: specular transmittance
    float on_a, on_b;  // matte with sigma != 0: Oren-Nayar A, B (on_b == 0 and on_a == 1 otherwise)
    int kd_tex, ks_tex, kr_tex, kt_tex;  // image texture replacing the constant at a hit, or -1
    int bump_tex;                        // float image texture displacing the shading geometry (Material::Bump), or -1
    int sigma_tex;                       // float image texture for matte's "sigma" (Oren-Nayar A, B per hit), or -1
    int rough_tex, remap_roughness;      // float image texture for "roughness" (-1: the constant alpha), RoughnessToAlpha or not
    float opacity[3];                    // uber: "opacity" ({1, 1, 1} otherwise)
    float alpha_y;                       // uber, glass: the distribution's alpha along v ("vroughness"); = alpha otherwise
    int opacity_tex;                     // uber: image texture for "opacity" (times the constant), or -1
    int rough_tex_v;                     // uber: -1 alpha_y is the constant, -2 alpha_y = the hit's alpha, >= 0 float image for "vroughness"
};
// ImageTexture + MIPMap (iile_texture): level l holds w x h float4 texels (rgb, w unused) at
// texels[offset[l] + t * w + s], row 0 = bottom scanline
constexpr int kMaxTexLevels = 16;
enum { kWrapRepeat = 0, kWrapBlack = 1, kWrapClamp = 2 };
struct DTexture {
    int n_levels, wrap, trilinear;
    float max_aniso, su, sv, du, dv;
    int level_w[kMaxTexLevels], level_h[kMaxTexLevels];
    long long level_offset[kMaxTexLevels];
};
enum { kLightDiffuseArea = 0, kLightPoint = 1, kLightSpot = 2, kLightDistant = 3, kLightAreaTriangle = 4,
       kLightInfinite = 5 };  // = IILE_LIGHT_* (checked in api.hip)
struct DLight {
    float lemit[3];  // area: Lemit; point: I
    int two_sided;
    int sphere;
    int type;        // kLight*
    float pos[3];    // point, spot: pLight; distant: wLight
    float w2l[9];    // spot: upper 3x3 of WorldToLight, row major
    float cos_total_width, cos_falloff_start, world_radius;
    int prim;        // triangle emitter: its primitive
    // infinite light (no map): LightToWorld 3x3 and the 2 x 2 Distribution2D, see iile_scene.h
    float l2w[9];
    int env_tex;            // infinite: Lmap among the textures
    int dist_w, dist_h;     // infinite: size of the Distribution2D
    long long dist_offset;  // its tables in DScene::env_dist
};
// Per Halton dimension: base, float reciprocal and offset of its digit permutation.
// The digits are peeled in double arithmetic (exact for any u32 index, see
// scrambled_radical_inverse): FP64 is full rate on CDNA, 32-bit integer multiplies are not.
struct DHaltonDim {
    uint32_t base, perm_offset;
    float inv_base;
    float perm0_term;  // invBase * perm[0] / (1 - invBase), lowdiscrepancy.cpp:422
    double base_d;     // double(base)
    double inv_base_d; // 1.0 / base, only ever used to estimate a quotient that is then exact
};
constexpr int kMaxHaltonDims = 128;
constexpr int kMaxSpheres = 8;
constexpr int kMaxMaterials = 64;
constexpr int kMaxLights = 8;
constexpr int kLightDistStride = 2 * kMaxLights + 2;  // a light distribution: func[kMaxLights], cdf[kMaxLights + 1], funcInt

// One IISPT probe camera (HemisphericCamera, hemispheric.cpp:109-160): CameraToWorld and the rows that transform a
// world normal into the camera's frame (transpose of WorldToCamera's mInv, transform.h:243-249)
struct DProbeCam {
    M44 c2w;
    float nrm[9];
};

#ifndef IILE_TOP_RECORDS
#define IILE_TOP_RECORDS 21  // levels 0..2 of the four-wide tree (1 + 4 + 16); 85 = levels 0..3
#endif
constexpr int kMaxTop = IILE_TOP_RECORDS;  // records of the tree's top kept in LDS by the traversal kernels (dpath.h)
// The three split axes of a four-wide record ride in the low two bits of its first three refs (ref << kRefShift | axis) instead of
// a word of their own, so that k_extend's interior step issues 7 vector loads per lane instead of 8 — on the deep-tree room the
// traversal kernels retire vector-memory lane-loads at the rate the L1 path allows at all (tools/vmem_calib.hip,
// profiles/r04_vmem_calib.json), and the only way to go faster is fewer of them. (The code paths for kRefShift == 0 — an axes word
// of its own at byte 112 of the record — are what the packing test probe iile_bvh_pack_probe / iile_wide_ref_shift describe.)
constexpr int kRefShift = 2;
constexpr int kTopFlag = 1 << 28;  // reference to a record of the LDS top: kTopFlag | slot (survives the shift)

struct DScene {
    // HBM arrays
    const float4 *wide;       // 4 float4 per interior node: both child boxes + child refs + split axis
    const float4 *wide4;      // four-wide records (8 float4 per interior node), see trav_interior4
    const float4 *tri_verts;  // 3 float4 per primitive: (p.xyz, w): w0=flags w1=material w2=light
    const float4 *tri_norms;  // 3 float4 per primitive: (n.xyz, uv.{x,y} spread over w)
    const float2 *tri_uv;     // 3 float2 per primitive
    const int *prim_shape;    // sphere index for sphere primitives
    const int2 *prim_alpha;   // {alphaMask, shadowAlphaMask} texture per primitive (has_alpha scenes)
    const uint16_t *perms;
    const DHaltonDim *hdims;
    const uint32_t *pixel_offsets;  // [128*128] Halton index offset of pixel (x mod 128, y mod 128)
    const DSphere *spheres;
    const DMaterial *materials;
    const DLight *lights;
    const DTexture *textures;  // image textures (k_shade<.., TEX = true>)
    const float4 *texels;
    const float *ewa_lut;      // MIPMap::weightLut[128]
    const float *env_dist;     // Distribution2D tables of the infinite lights
    int n_textures;
    int textured_materials;    // some material reads an image texture
    int n_nodes, n_prims, n_spheres, n_materials, n_lights, n_hdims;
    int n_perms;              // u16 entries of `perms`
    float root_box[6];        // bounds of the root node (min.xyz, max.xyz)
    int root_ref;             // >= 0: wide record; < 0: ~first primitive of a single-leaf tree
    // The top levels of the four-wide tree once more (breadth first, at most kMaxTop records of 8 float4): every block of a
    // traversal kernel keeps a copy in LDS, and references between them are kTopFlag | slot (dpath.h, load_wide4).
    const float4 *top4;
    int n_top;
    int root_ref_top;         // root_ref, or kTopFlag | 0 when the root record is among them
    // SpatialLightDistribution (n_lights > 1): per voxel {func[kMaxLights], cdf[kMaxLights + 1], funcInt}
    const float *light_dist;
    int light_nv[3];
    int has_infinite;         // some light is an InfiniteAreaLight: escaped rays carry radiance (k_miss)
    int extended_features;    // anything beyond one emitting sphere + matte / plastic: k_shade<.., EXT = true>
    int all_lights_infinite;  // every light is an InfiniteAreaLight: k_mis walks unordered (kernels_trav.hip)
    int has_glass;            // some material transmits: the paths' etaScale is tracked
    int has_uber_trans;       // some uber material has a SpecularTransmission lobe (opacity < 1 or Kt): a vertex can hold two such lobes (k_direct_tree<.., 2>)
    int has_specular;         // some material has a specular lobe (mirror, glass, uber): emitted light after such a bounce
    int has_alpha;            // some mesh has an alpha mask: the ALPHA builds of the traversal kernels run
    int boxes_nested;         // every child box lies inside its parent's (checked at upload): the four-wide
                              // step's skipping of intermediate nodes is exact only then
    // camera
    M44 raster_to_camera, camera_to_world;
    float lens_radius, focal_distance;
    float dx_camera[3], dy_camera[3];  // PerspectiveCamera::dxCamera / dyCamera
    float diff_scale;                  // 1 / sqrt(spp): ScaleDifferentials of the render loop
    // film
    int xres, yres;
    int crop_x0, crop_y0, crop_x1, crop_y1;
    int samp_x0, samp_y0, samp_x1, samp_y1;
    int pb_x0, pb_y0, pb_x1, pb_y1;   // the integrator's pixelBounds ("pixelbounds"): inside the sample bounds; pixels outside take no samples
    int pb_set;                       // 1: the pixel bounds are smaller than the sample bounds
    float filter_rx, filter_ry, max_sample_luminance;
    int probe_mode;              // IISPT probe pass: hemispheric cameras, IISPTdIntegrator::Li (k_shade / k_miss)
    int filter_wide;             // not the one-pixel box: samples are kept and gathered (k_film_store / k_film_gather)
    const float *filter_table;   // Film::filterTable, 16 x 16
    // halton
    int base_scale0, base_scale1, base_exp0, base_exp1, sample_stride, mult_inv0, mult_inv1;
    int sample_center;  // dimensions 0 and 1 of every sample are 0.5
    // sobol (iile_sobol): the frame's sampler is SobolSampler — generator matrices [n_dims][32], then vdc[32], vdc_inv[32]
    int sobol, sobol_log2res, sobol_res, sobol_dims;
    const uint32_t *sobol_mat;
    const uint32_t *sobol_vdc;
    // byte tables of the same matrices: entry [q][v] = XOR of the columns 8 q + j over the set bits j of the byte v, so a
    // 32-bit index takes four lookups instead of a loop over its bits. sobol_bt: [n_dims][4][256];
    // sobol_vdc_bt: [2][4][256] for the sample number k and for the interleaved pixel bits b of SobolIntervalToIndex
    const uint32_t *sobol_bt;
    const uint32_t *sobol_vdc_bt;
    // integrator
    int max_depth;
    float rr_threshold;
};

// queue records -------------------------------------------------------------
// ray:  ro = (o.xyz, bitcast pid)   rd = (d.xyz, tmax)
// hit:  (bitcast prim or -1, b0 | t, b1, b2)
// NEE record (five of the seven float4 planes of PassBuffers::nee, indexed by record slot):
//   n0 = (shadow o.xyz, light selection pdf)   n1 = (shadow d.xyz, bitcast flags)
//   n4 = (A.xyz, bitcast flags)   n5 = (B.xyz, bitcast light)   n6 = (beta.xyz, bitcast pid)
// (flags / light / pid are repeated so that each consumer streams only the planes it needs)
// MIS rays (planes 2 and 3, a dense queue of their own, indexed by MIS-queue slot: only records whose
// BSDF-sampled ray can matter have one):
//   n2 = (mis o.xyz, bitcast record slot)   n3 = (mis d.xyz, bitcast light)
// plus a byte plane written by k_mis / k_mis_lit, indexed by record slot:
// nee_mis (area light index + 1 of the primitive the MIS ray ended on, 0 = none; then 1 = lit)
enum { NEE_HAS_SHADOW = 1, NEE_HAS_MIS = 2 };

struct DCounters {
    unsigned long long camera_rays, closest_rays, shadow_rays;
    unsigned long long nodes_closest, nodes_any, tri_tests, tri_hits, sphere_tests;
    unsigned long long nee_evals, zero_radiance;
    unsigned long long path_length[8];
    // the extend kernel alone (closest-hit rays of the main path), for its roofline
    unsigned long long ext_rays, ext_nodes, ext_tri_tests, ext_sphere_tests;
    unsigned long long any_tri_tests;  // triangle tests of the shadow kernel
    // counted by every build (one atomic per wavefront at kernel end): the MIS rays k_mis really traced — the
    // uninstrumented k_shade drops those that cannot reach the sampled light
    unsigned long long mis_traced;
    // likewise the extension rays k_extend really traced (the uninstrumented pass of a scene without specular lobes does
    // not trace the rays of bounce maxDepth: path.cpp:104 breaks right after that intersection)
    unsigned long long ext_traced;
};

}  // namespace iile
