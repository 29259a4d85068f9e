// How much static LDS one workgroup may hold on gfx950 (the CU has 160 KB): launches kernels with 64 .. 160 KB and checks a
// write / read at both ends.  hipcc --offload-arch=gfx950 tools/lds_probe.hip -o tools/_build/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB>
__global__ void k(int *out) {
    __shared__ int buf[KB * 256];
    for (int i = threadIdx.x; i < KB * 256; i += blockDim.x) buf[i] = i ^ 0x5a5a;
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < KB * 256; i += blockDim.x) bad += buf[KB * 256 - 1 - i] != ((KB * 256 - 1 - i) ^ 0x5a5a);
    __syncthreads();
    // atomics at the far end of the allocation: every thread adds 1 to the last word and takes a ticket from the one before
    if (threadIdx.x == 0) buf[KB * 256 - 1] = 0, buf[KB * 256 - 2] = 0;
    __syncthreads();
    atomicAdd(&buf[KB * 256 - 1], 1);
    const int ticket = atomicAdd(&buf[KB * 256 - 2], 1);
    atomicMin((unsigned *)&buf[KB * 256 - 3], 7u);
    __syncthreads();
    if (threadIdx.x == 0) bad += (buf[KB * 256 - 1] != int(blockDim.x)) * 1000 + (buf[KB * 256 - 2] != int(blockDim.x)) * 10000 + (buf[KB * 256 - 3] != 7) * 100000;
    bad += (ticket < 0 || ticket >= int(blockDim.x)) * 1000000;
    atomicAdd(out, bad);
}
template <int KB>
void run(int *d) {
    hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k<KB>, dim3(1), dim3(512), 0, 0, d);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    int h = -1;
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%3d KB: launch %s, sync %s, mismatches %d\n", KB, hipGetErrorString(e), hipGetErrorString(e2), h);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    fflush(stdout);
    int *d;
    hipMalloc(&d, 4);
    run<64>(d);
    run<80>(d);
    run<96>(d);
    run<128>(d);
    run<140>(d);
    run<160>(d);
    return 0;
}
