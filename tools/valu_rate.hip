// Micro-benchmark: per-SIMD issue rate of the integer VALU ops the mask kernel is built from.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(int* out, int n, int a, int b) {
    int v0 = threadIdx.x, v1 = v0 + a, v2 = v0 ^ b, v3 = v0 * 3, v4 = v0 + 7, v5 = v0 - b, v6 = a - v0, v7 = v0 | b;
    for (int i = 0; i < n; ++i) {
#define STEP(x)                                                                                         \
    if (OP == 0) x = x + a;                                                                             \
    else if (OP == 1) x = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b), x, false); \
    else if (OP == 2) x = (((x << 8) >> 8) * ((a << 8) >> 8)) + b;                                           \
    else if (OP == 3) x = __builtin_amdgcn_perm(x, a, b);                                               \
    else if (OP == 4) x = __builtin_amdgcn_alignbit(x, a, 16);                                          \
    else if (OP == 5) x = x * a;                                                                        \
    else if (OP == 6) x = __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
#pragma unroll
        for (int u = 0; u < 8; ++u) { STEP(v0) STEP(v1) STEP(v2) STEP(v3) STEP(v4) STEP(v5) STEP(v6) STEP(v7) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}
template <int OP>
void run(const char* name, int* d) {
    const int n = 2000, blocks = 256 * 8, threads = 256;  // 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 10, 3, 5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, n, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * (threads / 64) * n * 64.0;  // wave-instructions
    printf("%-14s %8.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (insts / 1024.0));
}
int main() {
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_add_u32", d); run<1>("v_dot2c_i16", d); run<2>("v_mad_i32_i24", d); run<3>("v_perm_b32", d);
    run<4>("v_alignbit", d); run<5>("v_mul_lo_u32", d); run<6>("v_mov_dpp", d);
    return 0;
}
