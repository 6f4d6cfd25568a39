// Device-side helpers shared by the kernel translation units (wfa_kernels.hip, wfa_stream.hip).
// Everything sits in an anonymous namespace: each TU gets its own inlined copy.
#pragma once

#include "wfa_kernels.hpp"

namespace wfa {

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
#ifndef WFA_SPAN_WAVES
#define WFA_SPAN_WAVES 4  // waves per SIMD the span kernel is register-budgeted for (6 and 8 measured slower)
#endif

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_in_block() { return threadIdx.x >> 6; }

__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double uniform_f64(double v) {
    return __longlong_as_double(uniform_i64(__double_as_longlong(v)));
}

// ---- wave-level reductions (all 64 lanes participate) ---------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, kWave));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, kWave));
    return v;
}
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
// first-occurrence argmax: larger value wins, ties go to the smaller index (np.argmax).
__device__ __forceinline__ void wave_argmax(double& v, int& idx) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double ov = __shfl_xor(v, m, kWave);
        int oi = __shfl_xor(idx, m, kWave);
        bool take = (ov > v) || (ov == v && oi < idx);
        v = take ? ov : v;
        idx = take ? oi : idx;
    }
}

// ---- Savitzky-Golay, literal float64 evaluation ---------------------------------------------
// Interior: scipy.ndimage.correlate1d as called by savgol_filter(mode="interp") (reference call
// site cpu/filtering.py:234-240).  Symmetric branch:  tmp = x[c]*fw[h];
// for jj=-h..-1: tmp += (x[c+jj] + x[c-jj]) * fw[h+jj];  general branch: tmp = x[c+h]*fw[w-1];
// for jj=-h..h-1: tmp += x[c+jj]*fw[h+jj].  Edges: the degree-P least-squares polynomial of the
// first / last w samples evaluated at the edge positions (scipy _fit_edges_polyfit), here as a
// precomputed projection row.  Result rounded to float32 like scipy's float32 output array.
struct SgView {
    int w;         // effective window (0 = copy)
    int h;         // w / 2
    int sym;       // correlate1d symmetric branch?
    const double* fw;
    const double* el;  // [h][W] rows
    const double* er;
    int row;  // row stride (= plan W)
};

__device__ __forceinline__ SgView sg_view(const SgParams& sg, int L) {
    SgView v;
    int w = sg.W < L ? sg.W : L;
    if ((w & 1) == 0) w -= 1;
    if (w <= sg.P || w <= 0) {  // cpu/filtering.py:181-195: no-op filter
        v.w = 0; v.h = 0; v.sym = 1; v.fw = nullptr; v.el = nullptr; v.er = nullptr; v.row = sg.W;
        return v;
    }
    int t = (w - 1) >> 1;
    const double* base = sg.tab + (int64_t)t * sg.stride;
    v.w = w; v.h = w >> 1; v.sym = sg.sym[t];
    v.fw = base; v.el = base + sg.W; v.er = base + sg.W + sg.H * sg.W; v.row = sg.W;
    return v;
}

__device__ __forceinline__ float sg_value_f64(const uint16_t* __restrict__ x, int L, int i,
                                              const SgView& v) {
    if (v.w == 0) return (float)x[i];
    const int h = v.h, w = v.w;
    if (i < h) {
        const double* e = v.el + i * v.row;
        double s = 0.0;
        for (int k = 0; k < w; ++k) s += e[k] * (double)x[k];
        return (float)s;
    }
    if (i >= L - h) {
        const double* e = v.er + (i - (L - h)) * v.row;
        const uint16_t* xx = x + (L - w);
        double s = 0.0;
        for (int k = 0; k < w; ++k) s += e[k] * (double)xx[k];
        return (float)s;
    }
    const double* fw = v.fw;
    double tmp;
    if (v.sym) {
        tmp = (double)x[i] * fw[h];
        for (int jj = -h; jj < 0; ++jj)
            tmp += ((double)x[i + jj] + (double)x[i - jj]) * fw[h + jj];
    } else {
        tmp = (double)x[i + h] * fw[w - 1];
        for (int jj = -h; jj < h; ++jj) tmp += (double)x[i + jj] * fw[h + jj];
    }
    return (float)tmp;
}

// Wave value of sample i of a record, as float64, exactly as the reference consumer sees it:
// raw uint16 -> float64, filtered float32 -> float64 (records_view.py:229-253, dtype=float64).
template <int SRC>
struct WaveSrc {
    const uint16_t* xu;
    const float* xf;
    int L;
    SgView sg;
    __device__ __forceinline__ double at(int i) const {
        if (SRC == WFA_SRC_RAW) return (double)xu[i];
        if (SRC == WFA_SRC_F32) return (double)xf[i];
        return (double)sg_value_f64(xu, L, i, sg);
    }
    // float32 view (records_view.py:94: wave.astype(float32))
    __device__ __forceinline__ float at_f32(int i) const {
        if (SRC == WFA_SRC_RAW) return (float)xu[i];
        if (SRC == WFA_SRC_F32) return xf[i];
        return sg_value_f64(xu, L, i, sg);
    }
};

template <int SRC>
__device__ __forceinline__ WaveSrc<SRC> make_src(const PoolView& pool, const SgParams& sg,
                                                 int64_t off, int L) {
    WaveSrc<SRC> s;
    s.xu = pool.u16 ? pool.u16 + off : nullptr;
    s.xf = pool.f32 ? pool.f32 + off : nullptr;
    s.L = L;
    if (SRC == WFA_SRC_SG_FUSED) s.sg = sg_view(sg, L);
    return s;
}

// ---- row writers (packed little-endian rows, 4-byte aligned) --------------------------------
__device__ __forceinline__ void put_i64(uint32_t* row, int dword, int64_t v) {
    row[dword] = (uint32_t)(uint64_t)v;
    row[dword + 1] = (uint32_t)((uint64_t)v >> 32);
}
__device__ __forceinline__ void put_f32(uint32_t* row, int dword, float v) {
    row[dword] = __float_as_uint(v);
}
__device__ __forceinline__ void put_f64(uint32_t* row, int dword, double v) {
    put_i64(row, dword, __double_as_longlong(v));
}


// ---- exact-integer Savitzky-Golay building blocks (see wfa_kernels.hip, "K7 fast path") -----------------------
typedef short wfa_s2 __attribute__((ext_vector_type(2)));

typedef unsigned short wfa_us2 __attribute__((ext_vector_type(2)));
// acc + lo(pair) * lo(coef) + hi(pair) * hi(coef) on unsigned 16-bit halves
__device__ __forceinline__ uint32_t udot2_acc(uint32_t pair, uint32_t coef, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(wfa_us2, pair), __builtin_bit_cast(wfa_us2, coef), acc, false);
}
__device__ __forceinline__ int sdot2_acc(uint32_t pair, uint32_t coef, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(wfa_s2, pair), __builtin_bit_cast(wfa_s2, coef), acc, false);
}
// first tap of an accumulator: VOP3P form with an inline 0 addend (hipcc otherwise emits v_mov 0 + v_dot2c)
__device__ __forceinline__ int sdot2_first(uint32_t pair, uint32_t coef) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(pair), "v"(coef));
    return r;
}
__device__ __forceinline__ uint32_t dpp_from_prev_lane(uint32_t lane0_value, uint32_t v) {
    // lane i <- lane i-1 ; lane 0 keeps lane0_value (wave_shr:1, bound_ctrl off)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0_value, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t dpp_from_next_lane(uint32_t lane63_value, uint32_t v) {
    // lane i <- lane i+1 ; lane 63 keeps lane63_value (wave_shl:1)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane63_value, (int)v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
// Sum over the wave with DPP row shifts / broadcasts only (no LDS crossbar); result is wave-uniform.
__device__ __forceinline__ int wave_sum_i32_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8  -> lane 15 of each row = row sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast:31 into rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_excl_scan_i32(int v, int& total) {
    // inclusive scan over the 64 lanes with DPP row shifts / broadcasts, returned as exclusive
    int s = v;
    s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, true);  // row_shr:1
    s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, true);  // row_shr:2
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, true);  // row_shr:4
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, true);  // row_shr:8
    s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1, 3
    s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2, 3
    total = __builtin_amdgcn_readlane(s, 63);
    return s - v;
}

__device__ __forceinline__ int clamp_to_i32(double v) {
    return v >= 2147483647.0 ? INT32_MAX : (v <= -2147483648.0 ? INT32_MIN : (int)v);
}

// integer candidate band:  candidate <=> Z < zhi ; certainly masked <=> Z <= zlo
__device__ __forceinline__ void int_band(bool positive, double b, double thr, double den, double shift,
                                         int margin, int& zhi, int& zlo) {
    // negative/unknown: mask <=> y <= b - thr ; positive: mask <=> -y <= -(b + thr)
    const double tq = positive ? -(b + thr) * den : (b - thr) * den;
    if (!(tq == tq)) { zhi = INT32_MIN; zlo = INT32_MIN; return; }  // NaN: no hits
    const double tz = floor(tq) + shift;
    zhi = clamp_to_i32(tz + (double)margin + 1.0);
    zlo = clamp_to_i32(tz - (double)margin);
}

// ---- threshold-hit helpers shared by the hit kernels (hit_finder.py:231-255, 329-413) ------------------------------
struct HitCtx {
    double baseline;
    double thr;
    int positive;
    int L;
    int max_len;
    int le, re;
};

template <int SRC>
__device__ __forceinline__ double hit_signal(const WaveSrc<SRC>& src, const HitCtx& hc, int i) {
    // samples in [L, max_len) are the zero padding of the reference's dense matrix
    // (records_view.py:229-253); signal = w - b for "positive", b - w otherwise (hit_finder.py:240)
    const double w = (i < hc.L) ? src.at(i) : 0.0;
    return hc.positive ? (w - hc.baseline) : (hc.baseline - w);
}

// order-independent variant (first maximum by explicit index comparison)
struct HitAccAny {
    double best;
    int best_i;
    double sum;
    __device__ __forceinline__ void add(double s, int i) {
        if (s > best || (s == best && i < best_i)) { best = s; best_i = i; }
        sum += s > 0.0 ? s : 0.0;
    }
};

struct HitAcc {
    double best;
    int best_i;
    double sum;
    __device__ __forceinline__ void add(double s, int i) {
        if (s > best) { best = s; best_i = i; }  // ascending i: first maximum kept (np.argmax)
        sum += s > 0.0 ? s : 0.0;                 // hit_finder.py:380
    }
};

__device__ __forceinline__ void write_hit_row(uint8_t* __restrict__ out, int64_t h, const RecView& rec, int64_t r,
                                              int L, int start, int end, int seg_start, int seg_end,
                                              int pos, double best, double sum) {
    const int dt_ns = rec.dt[r];
    const double sip = (double)dt_ns * 1e3;  // hit_finder.py:382
    const int64_t rise = (int64_t)(pos - start > 0 ? pos - start : 0) * dt_ns;
    const int64_t fall = (int64_t)((end - 1) - pos > 0 ? (end - 1) - pos : 0) * dt_ns;
    const int64_t gts = (int64_t)((double)rec.ts[r] + (double)pos * sip);  // :383-386
    const int rl = L > 0 ? L : 0;
    int es = seg_start < rl ? seg_start : rl;
    int ee = seg_end < rl ? seg_end : rl;
    if (ee < es) ee = es;
    uint32_t* row = reinterpret_cast<uint32_t*>(out + h * 60);
    put_i64(row, 0, (int64_t)pos);
    put_f32(row, 2, (float)best);
    put_f32(row, 3, (float)sum);
    row[4] = (uint32_t)es;
    row[5] = (uint32_t)ee;
    put_f32(row, 6, (float)(double)(ee - es));
    row[7] = (uint32_t)dt_ns;
    put_f32(row, 8, (float)(double)rise);
    put_f32(row, 9, (float)(double)fall);
    put_i64(row, 10, gts);
    row[12] = (uint32_t)(uint16_t)rec.board[r] | ((uint32_t)(uint16_t)rec.chan[r] << 16);
    put_i64(row, 13, rec.rid[r]);
}

// 64-bit values across the 8 lanes of a group with DPP only
__device__ __forceinline__ double dpp_f64(double v, int ctrl_sel) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(uint32_t)(uint64_t)b, hi = (int)(uint32_t)((uint64_t)b >> 32);
    if (ctrl_sel == 0) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); }
    else if (ctrl_sel == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); }
    else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); }
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ __forceinline__ int dpp_i32(int v, int ctrl_sel) {
    if (ctrl_sel == 0) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    if (ctrl_sel == 1) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);                      // row_half_mirror
}

}  // namespace

}  // namespace wfa
