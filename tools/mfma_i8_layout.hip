// Probe: operand / result lane maps of v_mfma_i32_16x16x64_i8 on gfx950 with exact integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int8_t* A /*16x64 row-major*/, const int8_t* B /*64x16 row-major*/, int* D /*16x16*/) {
    const int l = threadIdx.x;
    v4i a, b, c = {0, 0, 0, 0};
    // hypothesis: lane l holds A[row l&15][k = 16*(l>>4) + j], B[k = 16*(l>>4) + j][col l&15], j = 0..15
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; ++j) { ab[j] = A[(l & 15) * 64 + 16 * (l >> 4) + j]; bb[j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)]; }
    __builtin_memcpy(&a, ab, 16); __builtin_memcpy(&b, bb, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    // hypothesis: D col = l&15, row = 4*(l>>4) + i
    for (int i = 0; i < 4; ++i) D[(4 * (l >> 4) + i) * 16 + (l & 15)] = c[i];
}
int main() {
    std::vector<int8_t> A(16 * 64), B(64 * 16);
    for (int i = 0; i < 16 * 64; ++i) A[i] = (int8_t)((i * 37 + 11) % 251 - 125);
    for (int i = 0; i < 64 * 16; ++i) B[i] = (int8_t)((i * 53 + 7) % 241 - 120);
    std::vector<int> ref(256, 0), got(256);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { int s = 0; for (int kk = 0; kk < 64; ++kk) s += (int)A[m * 64 + kk] * (int)B[kk * 16 + n]; ref[m * 16 + n] = s; }
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(got.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += got[i] != ref[i];
    printf("mfma_i32_16x16x64_i8 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
    return bad != 0;
}
