// Micro-benchmark: per-SIMD issue cost of the VALU / LDS-crossbar ops the streaming and row kernels are built from,
// at 2, 4 and 8 waves per SIMD.  Each wave runs 8 independent dependency chains of one op.
// Build: hipcc --offload-arch=gfx950 -O3 tools/op_rates.hip -o tools/op_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short s2 __attribute__((ext_vector_type(2)));

enum Op {
    ADD_U32, XOR_B32, CNDMASK, CMP_SGPR, DOT2_I16, DOT2_VOP3P, ALIGNBIT, MAX3_U32, MIN3_F32, MOV_DPP, MAD_I24, MUL_LO,
    PERM, ADD_F32, FMA_F32, ADD_F64, MUL_F64, MAX_F64, FMA_F64, CVT_F64_I32, CVT_F32_F64, CVT_F64_F32, CVT_F32_I32,
    CVT_I32_F32, BPERMUTE, READLANE, MBCNT, FFBL, LSHL_OR, ADD3, SAD_U16, UDOT2, N_OPS
};
static const char* kNames[N_OPS] = {
    "v_add_u32", "v_xor_b32", "v_cndmask_b32(vcc)", "v_cmp_lt_i32->sgpr + s_or", "v_dot2c_i32_i16", "v_dot2_i32_i16(vop3p)",
    "v_alignbit_b32", "v_max3_u32", "v_min3_f32", "v_mov_b32 dpp wave_shr", "v_mad_i32_i24", "v_mul_lo_u32", "v_perm_b32",
    "v_add_f32", "v_fma_f32", "v_add_f64", "v_mul_f64", "v_max_f64", "v_fma_f64", "v_cvt_f64_i32(+cvt back)",
    "v_cvt_f32_f64(+cvt back)", "v_cvt_f64_f32 pair", "v_cvt_f32_i32(+back)", "v_cvt_i32_f32 pair", "ds_bpermute_b32",
    "v_readlane+v_mov", "v_mbcnt lo+hi", "v_ffbl_b32", "v_lshl_or_b32", "v_add3_u32", "v_sad_u16", "v_dot2_u32_u16"};

template <int OP>
__device__ __forceinline__ void step(int& x, int a, int b, double& d, float& f, unsigned long long& sacc) {
    if (OP == ADD_U32) x = x + a;
    else if (OP == XOR_B32) x = x ^ a;
    else if (OP == CNDMASK) x = (x < b) ? a : x;  // cmp + cndmask: reported per pair
    else if (OP == CMP_SGPR) { sacc |= __ballot(x < a); x += 1; }
    else if (OP == DOT2_I16) x = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b), x, false);
    else if (OP == DOT2_VOP3P) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else if (OP == ALIGNBIT) x = __builtin_amdgcn_alignbit(x, a, 31);
    else if (OP == MAX3_U32) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if (OP == MIN3_F32) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f) : "v"(a), "v"(b));
    else if (OP == MOV_DPP) x = __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false);
    else if (OP == MAD_I24) x = (((x << 8) >> 8) * ((a << 8) >> 8)) + b;
    else if (OP == MUL_LO) x = x * a;
    else if (OP == PERM) x = __builtin_amdgcn_perm(x, a, b);
    else if (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(a));
    else if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(a), "v"(b));
    else if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d) : "v"(d));
    else if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(d));
    else if (OP == MAX_F64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d) : "v"(d));
    else if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d) : "v"(d));
    else if (OP == CVT_F64_I32) { asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(x)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(x) : "v"(d)); }
    else if (OP == CVT_F32_F64) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(d)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(f)); }
    else if (OP == CVT_F64_F32) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(f)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(d)); }
    else if (OP == CVT_F32_I32) { asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f) : "v"(x)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(x) : "v"(f)); }
    else if (OP == CVT_I32_F32) { asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(x) : "v"(f)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f) : "v"(x)); }
    else if (OP == BPERMUTE) x = __builtin_amdgcn_ds_bpermute(a, x);
    else if (OP == READLANE) x = __builtin_amdgcn_readlane(x, 5) + b;
    else if (OP == MBCNT) x = __builtin_amdgcn_mbcnt_hi(a, __builtin_amdgcn_mbcnt_lo(b, x));
    else if (OP == FFBL) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x));
    else if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(a));
    else if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if (OP == SAD_U16) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else if (OP == UDOT2) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
}

template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int n, int a, int b) {
    int v[8];
    double d[8];
    float f[8];
    unsigned long long sacc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = threadIdx.x * (u + 3) + a; d[u] = 1.0 + 1e-9 * (threadIdx.x + u); f[u] = 1.0f + 1e-3f * (threadIdx.x + u); }
    const int la = a + (threadIdx.x & 3) * 0, lb = b;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int u = 0; u < 8; ++u) step<OP>(v[u], la, lb, d[u], f[u], sacc);
        }
    }
    int r = (int)sacc;
#pragma unroll
    for (int u = 0; u < 8; ++u) r += v[u] + (int)d[u] + (int)f[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(int* dbuf, int waves_per_simd) {
    const int n = 1000, blocks = 256 * waves_per_simd, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, dbuf, 10, 3, 5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, dbuf, n, 3, 5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps_per_simd = (double)waves_per_simd * n * 32.0;  // step<> calls per SIMD
    printf("%-28s waves/SIMD %d  %8.3f ms  %6.2f cycles/step/SIMD @2.4GHz\n", kNames[OP], waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / steps_per_simd);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

template <int OP>
void run_all(int* d) {
    run<OP>(d, 2);
    run<OP>(d, 4);
    run<OP>(d, 8);
    if constexpr (OP + 1 < N_OPS) run_all<OP + 1>(d);
}

int main() {
    int* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run_all<0>(d);
    hipFree(d);
    return 0;
}
