// tools/vmem_calib.hip — what does a divergent, dependent record fetch cost on gfx950, per wave-INSTRUCTION or per LANE?
//
// The traversal kernels (k_extend / k_shadow) are bound by the vector-memory pipe in front of the L1 (profiles/r03a_pmc_mem.json).
// Whether fuller wavefronts would help them depends on how that pipe charges a load: per instruction (then a step with 26 of
// 64 lanes active wastes most of it and re-forming wavefronts pays) or per lane / per cache line touched (then lane utilisation is
// irrelevant and only fewer 16-byte pieces per step or lanes sharing lines can help). This tool measures it with the
// traversal's own access shape: every active lane chases its own chain through a table of 128-byte records (record i names the
// next record in piece 6, as a four-wide BVH record names its children there), fetching PIECES 16-byte pieces per step;
// 6 blocks of 256 threads per CU with 24 KB of LDS each, as the kernels run.
//   mode 0  per-lane records: lane l loads pieces {6, 0, 1, ..} of ITS record               (k_extend's interior step: 7 pieces)
//   mode 1  quad-cooperative: the 4 lanes of a quad share one chain; lane q loads pieces q and q + 4 of the quad's record
//           (2 instructions for the whole record, 16 chains per wavefront)
//   mode 2  as mode 0, but piece p comes from record (i + p * 977) mod n: PIECES different cache lines per step
//           (separates "per line" from "per instruction")
//   mode 3  mode 0 plus one 4-byte load from the same record; mode 4 / 5: the pieces as 12-byte / 4-byte loads
//           (is a load charged by the lane or by the byte?)
// Active lanes: the first n of the wavefront, or every (64 / n)-th.
// Build: hipcc -O3 --offload-arch=gfx950 tools/vmem_calib.hip -o tools/_build/vmem_calib ; run: vmem_calib [iters] > table.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

__device__ __forceinline__ uint32_t f2b(float f) { return __float_as_uint(f); }

template <int MODE, int PIECES>
__global__ __launch_bounds__(256) void k_chase(const float4 *__restrict__ table, uint32_t n_rec, int iters, unsigned long long active_mask,
                                               float *sink, unsigned long long *cycles) {
    extern __shared__ int lds_pad[];
    const int lane = threadIdx.x & 63;
    const bool active = (active_mask >> lane) & 1ull;
    // every chain starts somewhere else
    uint32_t idx = (uint32_t(blockIdx.x) * 256u + threadIdx.x) * 2654435761u % n_rec;
    if (MODE == 1) idx = (uint32_t(blockIdx.x) * 64u + (threadIdx.x >> 2)) * 2654435761u % n_rec;
    float acc = 0.f;
    if (threadIdx.x == 0) lds_pad[0] = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (active) {
        for (int it = 0; it < iters; ++it) {
            if (MODE == 0) {
                const float4 *r = table + size_t(idx) * 8;
                const float4 nx = r[6];
                float4 p[8];
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) p[k] = r[k < 6 ? k : 7];
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) acc += p[k].x + p[k].w;
                idx = f2b(nx.x);
            } else if (MODE == 1) {
                const float4 *r = table + size_t(idx) * 8 + (lane & 3);
                const float4 a = r[0], b = r[4];
                acc += a.x + a.w + b.y;
                // piece 6 sits in lane 2 of the quad (its second load): broadcast its .x to the quad
                const uint32_t nxt = f2b(b.x);
                idx = uint32_t(__builtin_amdgcn_mov_dpp(int(nxt), 0xAA, 0xf, 0xf, false));  // quad_perm [2,2,2,2]
            } else if (MODE == 3) {  // mode 0 + one more 4-byte load from the same record (k_extend's axes word)
                const float4 *r = table + size_t(idx) * 8;
                const float4 nx = r[6];
                const float meta = r[7].x;
                float4 p[8];
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) p[k] = r[k];
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) acc += p[k].x + p[k].w;
                acc += meta;
                idx = f2b(nx.x);
            } else if (MODE == 4) {  // the same pieces as 12-byte loads
                const float *r = reinterpret_cast<const float *>(table + size_t(idx) * 8);
                typedef float f3_t __attribute__((ext_vector_type(3)));
                const f3_t nx = *reinterpret_cast<const f3_t *>(r + 24);
                f3_t p[8];
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) p[k] = *reinterpret_cast<const f3_t *>(r + 4 * (k < 6 ? k : 7));
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) acc += p[k].x + p[k].z;
                idx = f2b(nx.x);
            } else if (MODE == 5) {  // ... as 4-byte loads
                const float *r = reinterpret_cast<const float *>(table + size_t(idx) * 8);
                const float nx = r[24];
                float p[8];
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) p[k] = r[4 * (k < 6 ? k : 7)];
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) acc += p[k];
                idx = f2b(nx);
            } else {
                const float4 nx = table[size_t(idx) * 8 + 6];
                float4 p[7];
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) {
                    uint32_t j = idx + uint32_t(k + 1) * 977u;
                    j = j >= n_rec ? j - n_rec : j;
                    p[k] = table[size_t(j) * 8 + (k < 6 ? k : 7)];
                }
#pragma unroll
                for (int k = 0; k < PIECES - 1; ++k) acc += p[k].x + p[k].w;
                idx = f2b(nx.x);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 12345.678f) sink[0] = acc + float(idx);
    if (lane == 0) atomicAdd(cycles, t1 - t0);
}

struct Row {
    int mode, pieces, n_active;
    const char *pattern;
    double table_mb;
    int blocks_per_cu;
    double ms, ns_per_step, steps_per_us_cu, lane_loads_per_ns_cu, memtime_per_step;
};

template <int MODE, int PIECES>
static Row run(const float4 *d_table, uint32_t n_rec, int iters, unsigned long long mask, const char *pattern, int n_cus, int blocks_per_cu,
               float *d_sink, unsigned long long *d_cyc) {
    const int grid = n_cus * blocks_per_cu;
    const size_t lds = 24 * 1024;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned long long cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(d_cyc, 0, 8));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_chase<MODE, PIECES>), dim3(grid), dim3(256), lds, 0, d_table, n_rec, iters, mask, d_sink, d_cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        }
    }
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    Row r;
    r.mode = MODE;
    r.pieces = MODE == 1 ? 8 : PIECES;
    r.n_active = __builtin_popcountll(mask);
    r.pattern = pattern;
    r.table_mb = double(n_rec) * 128 / 1048576.0;
    r.blocks_per_cu = blocks_per_cu;
    r.ms = best;
    const double waves_per_cu = blocks_per_cu * 4.0;
    r.ns_per_step = best * 1e6 / iters;                                       // one step of every wave (they run concurrently)
    const double chains_per_wave = MODE == 1 ? r.n_active / 4.0 : r.n_active;
    r.steps_per_us_cu = chains_per_wave * waves_per_cu * iters / (best * 1e3);  // record fetches per microsecond per CU
    const double loads_per_chain_step = MODE == 1 ? 8.0 : PIECES;
    r.lane_loads_per_ns_cu = r.steps_per_us_cu * loads_per_chain_step / 1e3;
    r.memtime_per_step = double(cyc) / (double(grid) * 4.0) / iters;
    return r;
}

static unsigned long long first_n(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }
static unsigned long long every(int n) {
    unsigned long long m = 0;
    for (int i = 0; i < n; ++i) m |= 1ull << (i * (64 / n));
    return m;
}
static unsigned long long quads_first(int nq) { return first_n(4 * nq); }

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    std::vector<Row> rows;
    float *d_sink;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_sink, 16));
    CHECK(hipMalloc(&d_cyc, 8));
    const double sizes_mb[3] = {1.0, 18.0, 72.0};  // L2-resident; the room's records; beyond one XCD's share of anything
    for (double mb : sizes_mb) {
        const uint32_t n_rec = uint32_t(mb * 1048576.0 / 128);
        // one random cycle through all records (Sattolo): next[i] in piece 6
        std::vector<uint32_t> perm(n_rec);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(12111);
        for (uint32_t i = n_rec - 1; i > 0; --i) std::swap(perm[i], perm[rng() % i]);
        std::vector<float4> h(size_t(n_rec) * 8);
        for (uint32_t i = 0; i < n_rec; ++i) {
            for (int p = 0; p < 8; ++p) h[size_t(i) * 8 + p] = make_float4(float(i & 255) * 1e-3f, 1.f, 2.f, float(p));
            float nx;
            memcpy(&nx, &perm[i], 4);
            h[size_t(i) * 8 + 6].x = nx;
        }
        float4 *d_table;
        CHECK(hipMalloc(&d_table, h.size() * sizeof(float4)));
        CHECK(hipMemcpy(d_table, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
        const int B = 6;
#define RUN(M, P, MASK, PAT) rows.push_back(run<M, P>(d_table, n_rec, iters, MASK, PAT, n_cus, B, d_sink, d_cyc))
        for (int n : {8, 16, 32, 64}) {
            RUN(0, 1, first_n(n), "first");
            RUN(0, 2, first_n(n), "first");
            RUN(0, 4, first_n(n), "first");
            RUN(0, 7, first_n(n), "first");
        }
        for (int n : {16, 32}) {
            RUN(0, 1, every(n), "every");
            RUN(0, 7, every(n), "every");
        }
        for (int nq : {4, 8, 16}) RUN(1, 8, quads_first(nq), "quads");
        for (int n : {16, 64}) {
            RUN(2, 4, first_n(n), "first");
            RUN(2, 7, first_n(n), "first");
        }
        // what a load costs by its width: 8 x 16 bytes, 7 x 16 + 4, 7 x 12, 7 x 4, with full and half-full wavefronts
        for (unsigned long long m : {first_n(64), every(32)}) {
            const char *pat = m == first_n(64) ? "first" : "every";
            RUN(0, 8, m, pat);
            RUN(3, 7, m, pat);
            RUN(4, 7, m, pat);
            RUN(5, 7, m, pat);
            RUN(0, 3, m, pat);
            RUN(4, 3, m, pat);
        }
        // fewer resident waves: 3 blocks per CU
        rows.push_back(run<0, 7>(d_table, n_rec, iters, first_n(64), "first", n_cus, 3, d_sink, d_cyc));
        rows.push_back(run<0, 7>(d_table, n_rec, iters, first_n(32), "first", n_cus, 3, d_sink, d_cyc));
        rows.push_back(run<1, 8>(d_table, n_rec, iters, quads_first(16), "quads", n_cus, 3, d_sink, d_cyc));
        CHECK(hipFree(d_table));
    }
    printf("{\n \"note\": \"tools/vmem_calib.hip: dependent fetches of 128-byte records, one chain per active lane (mode 0/2) or per quad (mode 1); "
           "%d steps per chain, 24 KB LDS per 256-thread block; steps_per_us_cu = record fetches per microsecond per CU, "
           "lane_loads_per_ns_cu = 16-byte lane-loads per nanosecond per CU, memtime_per_step = s_memtime ticks (100 MHz) per step of a wave\",\n"
           " \"device\": \"%s\", \"cus\": %d,\n \"rows\": [\n",
           iters, prop.gcnArchName, n_cus);
    for (size_t i = 0; i < rows.size(); ++i) {
        const Row &r = rows[i];
        printf("  {\"mode\": %d, \"pieces\": %d, \"active\": %d, \"pattern\": \"%s\", \"table_mb\": %.0f, \"blocks_per_cu\": %d, \"ms\": %.3f, "
               "\"ns_per_wave_step\": %.1f, \"steps_per_us_cu\": %.1f, \"lane_loads_per_ns_cu\": %.3f, \"memtime_per_step\": %.2f}%s\n",
               r.mode, r.pieces, r.n_active, r.pattern, r.table_mb, r.blocks_per_cu, r.ms, r.ns_per_step, r.steps_per_us_cu,
               r.lane_loads_per_ns_cu, r.memtime_per_step, i + 1 < rows.size() ? "," : "");
    }
    printf(" ]\n}\n");
    return 0;
}
