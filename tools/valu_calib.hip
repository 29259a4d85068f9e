// tools/valu_calib.hip — where does VALU issue saturate on gfx950, and what does each instruction class of k_shade cost?
//
// One kernel per (instruction, waves per SIMD): every wave runs ITERS iterations of 64 instructions of one kind on 8
// independent registers (no dependence closer than 8 instructions). K waves per SIMD on every SIMD of the chip: one block
// of 256 * K threads per CU for K <= 4 (its waves are dealt round-robin over the four SIMDs), two blocks of 128 * K threads
// for K = 6, 8; dynamic LDS (100 KB resp. 70 KB per block) keeps any further block off the CU and the grid is exactly what
// is resident. (A first version used 256-thread blocks and LDS slices of 160 KB / K: at K >= 6 fewer blocks than intended
// were resident, which the in-kernel stamps showed as 0.8 "instructions per cycle".) Each block stamps s_memtime around its loop: cycles per wave-instruction as the wave sees them, and
// K / that = wave-instructions per cycle per SIMD. The same binary under
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace
// gives the counters the `valu_busy` figure of tools/summarize_profiles.py is made of (tools/valu_calib.sh).
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_calib.hip -o gpurun_out/valu_calib
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                             \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

enum Op {
    FMA_F32, PK_FMA_F32, FMA_F64, RCP_F32, SQRT_F32, DIV_SCALE_F32, DIV_FMAS_F32, DIV_FIXUP_F32, MUL_LO_U32, MAD_U64_U32,
    CNDMASK, CMP_F32, MAX3_F32, RCP_F64, DS_READ_B32, IEEE_DIV_F32, IEEE_SQRT_F32, MIX_SHADE, N_OPS
};
static const char *kOpName[N_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_rcp_f32", "v_sqrt_f32", "v_div_scale_f32 (+ the s_nop the compiler pads behind an SGPR-writing VALU op)",
                                     "v_div_fmas_f32", "v_div_fixup_f32", "v_mul_lo_u32", "v_mad_u64_u32 (+ s_nop)", "v_cndmask_b32",
                                     "v_cmp_lt_f32 (+ s_nop)", "v_max3_f32", "v_rcp_f64", "ds_read_b32", "ieee_div_f32(a/b, correctly rounded)",
                                     "ieee_sqrt_f32(correctly rounded)", "mix 10 fma + 1 div + 1 sqrt per 12 ops"};
// wave-instructions per inner step of 8 (for the compiled sequences the disassembly count is printed by the .sh)
template <int OP>
__device__ __forceinline__ void step8(float (&a)[8], double (&d)[4], float x, float y, uint32_t (&u)[8], __attribute__((address_space(3))) int *lds,
                                      unsigned long long cond) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        if (OP == PK_FMA_F32) {
            if (i < 4) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 3]));
            else asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d[i - 4]) : "v"(d[(i + 1) & 3]));
        }
        if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i & 3]) : "v"(d[(i + 1) & 3]));
        if (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (OP == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        // (condition outputs go to an SGPR pair of the compiler's choice, not VCC: with a VCC clobber the compiler pads an
        //  s_nop behind every instruction and the row measures the padding)
        if (OP == DIV_SCALE_F32) {
            unsigned long long sc;
            asm volatile("v_div_scale_f32 %0, %1, %2, %3, %2" : "+v"(a[i]), "=s"(sc) : "v"(x), "v"(y));
        }
        if (OP == DIV_FMAS_F32) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));  // reads VCC as it stands
        if (OP == DIV_FIXUP_F32) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (OP == MAD_U64_U32) {
            unsigned long long sc;
            asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(d[i & 3]), "=s"(sc) : "v"(u[i]), "v"(u[(i + 1) & 7]));
        }
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "s"(cond));
        if (OP == CMP_F32) {
            unsigned long long sc;
            asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(sc) : "v"(a[i]), "v"(x));
        }
        if (OP == MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        if (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i & 3]));
        if (OP == DS_READ_B32) {
            int v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(lds));
            u[i] ^= uint32_t(v);
        }
        if (OP == IEEE_DIV_F32) a[i] = x / a[i];        // compiled with -fhip-fp32-correctly-rounded-divide-sqrt
        if (OP == IEEE_SQRT_F32) a[i] = __builtin_sqrtf(a[i]) + y;
        if (OP == MIX_SHADE) {
            if (i == 3) a[i] = x / a[i];
            else if (i == 7) a[i] = __builtin_sqrtf(a[i]);
            else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        }
    }
}

template <int OP, int K>
__global__ __launch_bounds__(K <= 4 ? 256 * K : 128 * K) void k_calib(float *out, unsigned long long *stamps, int iters, float x, float y) {
    extern __shared__ int s_lds[];
    float a[8];
    double d[4];
    uint32_t u[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.f + 0.001f * float(threadIdx.x + i), u[i] = threadIdx.x * 2654435761u + i;
    for (int i = 0; i < 4; ++i) d[i] = 1e-3 * double(threadIdx.x + i + 1);
    if (threadIdx.x < 64) s_lds[threadIdx.x] = int(threadIdx.x);
    __syncthreads();
    auto *lds = (__attribute__((address_space(3))) int *)(s_lds + (threadIdx.x & 63));
    const unsigned long long cond = __builtin_amdgcn_read_exec() & 0x5555555555555555ull;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) step8<OP>(a, d, x, y, u, lds, cond);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + float(u[i] & 1u);
    for (int i = 0; i < 4; ++i) s += float(d[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

struct Row {
    int op, k;
    double ms, cyc_per_inst, clock_ghz, inst_per_cyc_simd, wall_ops_per_cyc_simd = 0;
};

template <int OP, int K>
void run(float *out, unsigned long long *stamps, std::vector<Row> &rows, int iters) {
    const int n_cus = 256, blocks = K <= 4 ? n_cus : 2 * n_cus, threads = K <= 4 ? 256 * K : 128 * K, waves = blocks * threads / 64;
    const size_t lds = K <= 4 ? 100 * 1024 : 70 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_calib<OP, K>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    std::vector<unsigned long long> h(size_t(waves) * 2);
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_calib<OP, K>), dim3(blocks), dim3(threads), lds, 0, out, stamps, iters, 1.0001f, 0.9999f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int w = 0; w < waves; ++w) {
        cyc.push_back(double(h[2 * w]));
        clk.push_back(double(h[2 * w]) / (double(h[2 * w + 1]) * 10.0));  // s_memrealtime ticks at 100 MHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double insts = double(iters) * 64.0;
    Row r{OP, K, double(best), cyc[cyc.size() / 2] / insts, clk[clk.size() / 2], 0};
    r.inst_per_cyc_simd = double(K) / r.cyc_per_inst;
    r.wall_ops_per_cyc_simd = double(waves) * insts / 1024.0 / (double(best) * 1e-3 * r.clock_ghz * 1e9);
    rows.push_back(r);
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}
template <int OP>
void run_all_k(float *out, unsigned long long *stamps, std::vector<Row> &rows, int iters) {
    run<OP, 1>(out, stamps, rows, iters);
    run<OP, 2>(out, stamps, rows, iters);
    run<OP, 3>(out, stamps, rows, iters);
    run<OP, 4>(out, stamps, rows, iters);
    run<OP, 6>(out, stamps, rows, iters);
    run<OP, 8>(out, stamps, rows, iters);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    float *out;
    unsigned long long *stamps;
    CHECK(hipMalloc(&out, size_t(512) * 1024 * 4));
    CHECK(hipMalloc(&stamps, size_t(512) * 16 * 16));
    std::vector<Row> rows;
    run_all_k<FMA_F32>(out, stamps, rows, iters);
    run_all_k<PK_FMA_F32>(out, stamps, rows, iters);
    run_all_k<FMA_F64>(out, stamps, rows, iters);
    run_all_k<RCP_F32>(out, stamps, rows, iters);
    run_all_k<SQRT_F32>(out, stamps, rows, iters);
    run_all_k<DIV_SCALE_F32>(out, stamps, rows, iters);
    run_all_k<DIV_FMAS_F32>(out, stamps, rows, iters);
    run_all_k<DIV_FIXUP_F32>(out, stamps, rows, iters);
    run_all_k<MUL_LO_U32>(out, stamps, rows, iters);
    run_all_k<MAD_U64_U32>(out, stamps, rows, iters);
    run_all_k<CNDMASK>(out, stamps, rows, iters);
    run_all_k<CMP_F32>(out, stamps, rows, iters);
    run_all_k<MAX3_F32>(out, stamps, rows, iters);
    run_all_k<RCP_F64>(out, stamps, rows, iters);
    run_all_k<DS_READ_B32>(out, stamps, rows, iters);
    run_all_k<IEEE_DIV_F32>(out, stamps, rows, iters);
    run_all_k<IEEE_SQRT_F32>(out, stamps, rows, iters);
    run_all_k<MIX_SHADE>(out, stamps, rows, iters);
    printf("{\"iters\": %d, \"insts_per_iter\": 64, \"note\": \"cyc_per_op = median over waves of s_memtime delta / (iters * 64 source-level ops); "
           "ops_per_cyc_simd = waves_per_simd / cyc_per_op; for the compiled sequences (ieee_*, mix) an op is a whole a/b or sqrt expansion\", \"rows\": [\n",
           iters);
    for (size_t i = 0; i < rows.size(); ++i)
        printf(" {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cyc_per_op\": %.3f, \"ops_per_cyc_simd\": %.4f, \"ops_per_cyc_simd_by_wall_time\": %.4f, \"clock_ghz\": %.3f}%s\n",
               kOpName[rows[i].op], rows[i].k, rows[i].ms, rows[i].cyc_per_inst, rows[i].inst_per_cyc_simd, rows[i].wall_ops_per_cyc_simd, rows[i].clock_ghz,
               i + 1 < rows.size() ? "," : "");
    printf("]}\n");
    return 0;
}
