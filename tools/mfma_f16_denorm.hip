// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (gfx950)? The split-fp16 network (csrc/device/iispt_net.hip) stores the
// low halves of its operands in fp16, where values below 2^-14 are subnormal: if the matrix pipe flushed them the split would lose
// its low-order products for small operands. One wavefront, A = a constant c in every element, B = 1: every output = 16 c.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f16_denorm.hip -o tools/_build/mfma_f16_denorm && tools/_build/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(const float *vals, float *out, int n) {
    for (int i = 0; i < n; ++i) {
        half8 a, b;
        for (int j = 0; j < 8; ++j) a[j] = (_Float16)vals[i], b[j] = (_Float16)1.0f;
        f32x16 acc = {0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (threadIdx.x == 0) out[2 * i] = acc[0];
        // and the product of two subnormal-range pieces with a large partner: (c) x (1024)
        for (int j = 0; j < 8; ++j) b[j] = (_Float16)1024.0f;
        f32x16 acc2 = {0};
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
        if (threadIdx.x == 0) out[2 * i + 1] = acc2[0];
    }
}

int main() {
    const int n = 6;
    float h[n] = {1.0f, 6.103515625e-5f /* 2^-14: smallest normal */, 3.0517578125e-5f /* 2^-15 */, 5.9604644775390625e-8f /* 2^-24: smallest subnormal */,
                  1.5e-5f, 4.1e-6f};
    float *dv, *dout, out[2 * n];
    hipMalloc(&dv, sizeof(h));
    hipMalloc(&dout, sizeof(out));
    hipMemcpy(dv, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dv, dout, n);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    int flushed = 0;
    for (int i = 0; i < n; ++i) {
        const float c = float((_Float16)h[i]);
        printf("{\"input\": %.10g, \"as_f16\": %.10g, \"mfma_sum16\": %.10g, \"expect\": %.10g, \"times1024\": %.10g, \"expect1024\": %.10g}\n", h[i], c, out[2 * i], 16.f * c,
               out[2 * i + 1], 16.f * 1024.f * c);
        if (out[2 * i] != 16.f * c || out[2 * i + 1] != 16384.f * c) flushed = 1;
    }
    printf("{\"f16_subnormal_inputs_kept_by_mfma\": %s}\n", flushed ? "false" : "true");
    return 0;
}
