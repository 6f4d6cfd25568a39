// Micro-benchmark: issue rate of v_mfma_i32_16x16x64_i8 (4 independent accumulator chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int n, v4i a, v4i b) {
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    a[0] += threadIdx.x; b[1] ^= threadIdx.x;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    v4i a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int n = 2000, blocks = 256 * waves, threads = 256;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 10, a, b);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, n, a, b);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)blocks * 4 * n * 16.0 / 1024.0;
        printf("%d wave(s)/SIMD: %.3f ms, %.1f cycles per MFMA per SIMD @2.4GHz\n", waves, ms, ms * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
