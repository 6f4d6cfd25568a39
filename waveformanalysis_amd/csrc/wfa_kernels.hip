// HIP kernels (gfx950 / CDNA4) for the per-record waveform hot path.
//
// Every kernel works on the device-resident pool + records SoA of a wfa_ctx.  Work unit = one
// record per 64-lane wavefront (records are independent: SURVEY.md section 8e); blocks of 4
// waves grid-stride over the records.  Arithmetic follows the reference literally (float64
// where the reference computes in float64, float32 where it rounds to float32); the file is
// compiled with -ffp-contract=off so no multiply-add is fused that the reference does not fuse.
//
// Reference lines are cited at each device function (paths relative to waveform_analysis/).

#include "wfa_kernels.hpp"

namespace wfa {

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

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

}  // namespace

// =============================================================================================
// K1: baseline mean  (records_builder.py:243-257)
// =============================================================================================
__global__ __launch_bounds__(kBlock) void k_baseline_mean(PoolView pool, RecView rec, int32_t start,
                                                          int32_t end, double* __restrict__ out) {
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t r = wave0; r < rec.R; r += nwaves) {
        const int L = rec.len[r];
        const uint16_t* x = pool.u16 + rec.off[r];
        int e = end < L ? end : L;
        int64_t s = 0;
        for (int i = start + lane; i < e; i += kWave) s += x[i];
        s = wave_sum_i64(s);
        if (lane == 0) {
            // integer sum is exact, so this equals np.mean over float64 of the samples
            out[r] = (e <= start) ? __longlong_as_double(0x7ff8000000000000LL)
                                  : (double)s / (double)(e - start);
        }
    }
}

// =============================================================================================
// K2: wave_pool_filtered materialisation  (records.py:368-438, filtering.py:377-407)
// =============================================================================================
__global__ __launch_bounds__(kBlock) void k_savgol(PoolView pool, RecView rec, SgParams sg,
                                                   float* __restrict__ out) {
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t r = wave0; r < rec.R; r += nwaves) {
        const int L = rec.len[r];
        if (L <= 0) continue;
        const int64_t off = rec.off[r];
        const uint16_t* x = pool.u16 + off;
        SgView v = sg_view(sg, L);
        for (int i = lane; i < L; i += kWave) out[off + i] = sg_value_f64(x, L, i, v);
    }
}

// =============================================================================================
// K4 / K7: threshold hits  (hit_finder.py:231-255, 329-413)
// =============================================================================================
// Phase A builds the record's threshold mask as 64-bit ballot words in LDS; phase B walks the
// runs [start, end) with scalar bit scans and evaluates each hit window wave-cooperatively.
// Hit rows go to chunked temporary storage (one atomic per chunk, not per record); a scan over
// the per-record counts and a gather kernel then produce the (record, start)-ordered output.

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

template <int SRC, bool FUSED_BASELINE>
__global__ __launch_bounds__(kBlock) void k_hits(PoolView pool, RecView rec, SgParams sg,
                                                 HitParams hp) {
    extern __shared__ uint64_t lds_bm[];
    const int lane = lane_id();
    const int wv = wave_in_block();
    uint64_t* bm = lds_bm + (size_t)wv * hp.bm_words;
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wv;
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;

    // chunk allocator state (wave-uniform)
    int64_t chunk_base = 0;
    int64_t chunk_left = 0;

    for (int64_t r = wave0; r < rec.R; r += nwaves) {
        const int L = rec.len[r];
        const int64_t off = rec.off[r];
        WaveSrc<SRC> src = make_src<SRC>(pool, sg, off, L);

        HitCtx hc;
        hc.L = L; hc.max_len = hp.max_len; hc.le = hp.le; hc.re = hp.re;
        hc.thr = rec.thr[r];
        hc.positive = rec.pol[r] == WFA_POL_POSITIVE;
        if (FUSED_BASELINE) {
            // K1 fused: mean of raw samples [bl_start, min(bl_end, L))
            int e = hp.bl_end < L ? hp.bl_end : L;
            int64_t s = 0;
            for (int i = hp.bl_start + lane; i < e; i += kWave) s += src.xu[i];
            s = wave_sum_i64(s);
            hc.baseline = (e <= hp.bl_start) ? __longlong_as_double(0x7ff8000000000000LL)
                                             : (double)s / (double)(e - hp.bl_start);
            if (lane == 0) rec.baseline_rw[r] = hc.baseline;
        } else {
            hc.baseline = rec.baseline[r];
        }

        // ---- phase A: mask words + run count -------------------------------------------------
        const int nw = (L + kWave - 1) / kWave;
        int n_runs = 0;
        uint64_t prev_msb = 0;
        for (int wi = 0; wi < nw; ++wi) {
            const int i = wi * kWave + lane;
            bool m = false;
            if (i < L) {
                const double w = src.at(i);
                const double sig = hc.positive ? (w - hc.baseline) : (hc.baseline - w);
                m = sig >= hc.thr;  // hit_finder.py:346
            }
            const uint64_t word = __ballot(m);
            if (lane == 0) bm[wi] = word;
            const uint64_t starts = word & ~((word << 1) | prev_msb);
            n_runs += __popcll(starts);
            prev_msb = word >> 63;
        }
        n_runs = uniform_i32(n_runs);

        // ---- allocate rows ---------------------------------------------------------------------
        int64_t tmp_start = 0;
        if (n_runs > 0) {
            if (chunk_left < n_runs) {
                const int64_t need = n_runs > hp.chunk_rows ? n_runs : hp.chunk_rows;
                unsigned long long got = 0;
                if (lane == 0) got = atomicAdd(hp.cursor, (unsigned long long)need);
                chunk_base = uniform_i64((int64_t)got);
                chunk_left = need;
            }
            tmp_start = chunk_base;
            chunk_base += n_runs;
            chunk_left -= n_runs;
        }
        if (lane == 0) {
            hp.rec_tmp_start[r] = tmp_start;
            hp.rec_nhits[r] = n_runs;
        }
        if (n_runs == 0) continue;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();  // bm[] written by lane 0, read by all lanes below

        // ---- phase B: walk runs ----------------------------------------------------------------
        const int64_t ts = rec.ts[r];
        const int dt_ns = rec.dt[r];
        const double sip = (double)dt_ns * 1e3;  // hit_finder.py:382
        int cur = 0;
        for (int k = 0; k < n_runs; ++k) {
            // next set bit >= cur
            int wi = cur >> 6;
            uint64_t w = bm[wi] & (~0ull << (cur & 63));
            while (w == 0) w = bm[++wi];
            const int start = uniform_i32(wi * 64 + __ffsll((long long)w) - 1);
            // next clear bit > start
            int wj = start >> 6;
            uint64_t z = ~bm[wj] & (~0ull << (start & 63));
            while (z == 0 && ++wj < nw) z = ~bm[wj];
            const int end = uniform_i32(wj < nw ? wj * 64 + __ffsll((long long)z) - 1 : nw * 64);
            cur = end;

            const int seg_start = start - hc.le > 0 ? start - hc.le : 0;
            const int seg_end = end + hc.re < hc.max_len ? end + hc.re : hc.max_len;

            double best = -__builtin_huge_val();
            int best_i = 0x7fffffff;
            double sum = 0.0;
            for (int base = seg_start; base < seg_end; base += kWave) {
                const int i = base + lane;
                if (i < seg_end) {
                    const double s = hit_signal<SRC>(src, hc, i);
                    if (s > best) { best = s; best_i = i; }  // ascending i: first max kept
                    sum += s > 0.0 ? s : 0.0;                 // hit_finder.py:380
                }
            }
            wave_argmax(best, best_i);
            sum = wave_sum(sum);

            if (lane == 0) {
                const int64_t row_idx = tmp_start + k;
                if (row_idx < hp.tmp_rows) {
                    uint32_t* row = reinterpret_cast<uint32_t*>(hp.tmp + row_idx * 60);
                    const int pos = best_i;
                    const int64_t rise = (int64_t)(pos - start > 0 ? pos - start : 0) * dt_ns;
                    const int64_t fall = (int64_t)((end - 1) - pos > 0 ? (end - 1) - pos : 0) * dt_ns;
                    const int64_t gts = (int64_t)((double)ts + (double)pos * sip);  // :383-386
                    const int rl = L > 0 ? L : 0;
                    int es = seg_start < rl ? seg_start : rl;
                    int ee = seg_end < rl ? seg_end : rl;
                    if (ee < es) ee = es;
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
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- exclusive scan of per-record hit counts (int32 -> int64 offsets) --------------------------
constexpr int kScanTile = 1024;  // records per scan block (256 threads x 4)

__global__ __launch_bounds__(kBlock) void k_scan_block_sums(const int32_t* __restrict__ counts,
                                                            int64_t n, int64_t* __restrict__ sums) {
    __shared__ int64_t part[kWavesPerBlock];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 4;
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < n) s += counts[base + j];
    s = wave_sum_i64(s);
    if (lane_id() == 0) part[wave_in_block()] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// single block: exclusive scan of the block sums in place, total appended at sums[n_blocks]
__global__ __launch_bounds__(kBlock) void k_scan_sums(int64_t* __restrict__ sums, int64_t n_blocks) {
    __shared__ int64_t wave_tot[kWavesPerBlock];
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_blocks; base += kBlock) {
        const int64_t i = base + threadIdx.x;
        int64_t v = i < n_blocks ? sums[i] : 0;
        int64_t inc = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            int64_t o = __shfl_up(inc, d, kWave);
            if (lane_id() >= d) inc += o;
        }
        if (lane_id() == kWave - 1) wave_tot[wave_in_block()] = inc;
        __syncthreads();
        int64_t wave_off = 0;
        for (int w = 0; w < wave_in_block(); ++w) wave_off += wave_tot[w];
        const int64_t carry = carry_s;
        if (i < n_blocks) sums[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = carry + wave_off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[n_blocks] = carry_s;
}

__global__ __launch_bounds__(kBlock) void k_scan_apply(const int32_t* __restrict__ counts, int64_t n,
                                                       const int64_t* __restrict__ sums,
                                                       int64_t* __restrict__ out) {
    __shared__ int64_t wave_tot[kWavesPerBlock];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 4;
    int32_t c[4];
    int64_t s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = base + j < n ? counts[base + j] : 0;
        s += c[j];
    }
    int64_t inc = s;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        int64_t o = __shfl_up(inc, d, kWave);
        if (lane_id() >= d) inc += o;
    }
    if (lane_id() == kWave - 1) wave_tot[wave_in_block()] = inc;
    __syncthreads();
    int64_t run = sums[blockIdx.x] + inc - s;
    for (int w = 0; w < wave_in_block(); ++w) run += wave_tot[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (base + j < n) out[base + j] = run;
        run += c[j];
    }
}

// one thread per record: move its rows from chunk order to (record, start) order
__global__ __launch_bounds__(kBlock) void k_hits_gather(const uint8_t* __restrict__ tmp,
                                                        const int64_t* __restrict__ tmp_start,
                                                        const int32_t* __restrict__ nhits,
                                                        const int64_t* __restrict__ out_start,
                                                        int64_t R, uint8_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    const int n = nhits[r];
    if (n == 0) return;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(tmp + tmp_start[r] * 60);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + out_start[r] * 60);
    for (int k = 0; k < n * 15; ++k) dst[k] = src[k];
}

// =============================================================================================
// K5: basic features  (basic_features.py:108-195)
// =============================================================================================
__device__ __forceinline__ void py_slice(int64_t start, int64_t end, int has_end, int L, int& lo,
                                         int& hi) {
    int64_t s = start;
    if (s < 0) { s += L; if (s < 0) s = 0; } else if (s > L) s = L;
    int64_t e = has_end ? end : (int64_t)L;
    if (e < 0) { e += L; if (e < 0) e = 0; } else if (e > L) e = L;
    lo = (int)s;
    hi = (int)(e < s ? s : e);
}

template <int SRC>
__global__ __launch_bounds__(kBlock) void k_basic_features(PoolView pool, RecView rec, SgParams sg,
                                                           FeatParams fp, uint8_t* __restrict__ out) {
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const double inf = __builtin_huge_val();
    for (int64_t r = wave0; r < rec.R; r += nwaves) {
        const int L = rec.len[r];
        const int64_t off = rec.off[r];
        WaveSrc<SRC> src = make_src<SRC>(pool, sg, off, L);
        double baseline = rec.baseline[r];
        if (fp.fixed_bl) {
            const double fb = fp.fixed_bl[r];
            if (fb == fb) baseline = fb;  // basic_features.py:143-146
        }
        const int pol = rec.pol[r];
        const bool known = pol == WFA_POL_NEGATIVE || pol == WFA_POL_POSITIVE;
        int p0, p1, c0, c1;
        py_slice(fp.h0, fp.h1, fp.h_has_end, L, p0, p1);
        py_slice(fp.a0, fp.a1, fp.a_has_end, L, c0, c1);

        double vmin = inf, vmax = -inf, area = 0.0, mad = 0.0;
        if (known) {
            // s = -rv.signals(id, baseline): float32 arithmetic (records_view.py:87-100)
            const float b32 = (float)baseline;
            for (int i = lane; i < L; i += kWave) {
                const float d = src.at_f32(i) - b32;
                const double s = (double)(pol == WFA_POL_POSITIVE ? d : -d);
                if (i >= p0 && i < p1) { vmin = fmin(vmin, s); vmax = fmax(vmax, s); }
                if (i >= c0 && i < c1) area += s;
            }
        } else {
            for (int i = lane; i < L; i += kWave) {
                const double w = src.at(i);
                if (i >= p0 && i < p1) { vmin = fmin(vmin, w); vmax = fmax(vmax, w); }
                if (i >= c0 && i < c1) area += baseline - w;  // effective polarity "negative"
            }
        }
        // max |diff| over the whole record on the wave values (basic_features.py:187-189)
        for (int i = lane; i + 1 < L; i += kWave) {
            const double d = src.at(i + 1) - src.at(i);
            mad = fmax(mad, fabs(d));
        }
        vmin = wave_min(vmin); vmax = wave_max(vmax); area = wave_sum(area); mad = wave_max(mad);

        if (lane == 0) {
            float height = 0.f, amp = 0.f, area_f = 0.f, mad_f = 0.f;
            if (p1 > p0) {
                height = known ? (float)vmax : (float)(baseline - vmin);
                amp = (float)(vmax - vmin);
            }
            if (c1 > c0) area_f = (float)area;
            if (L > 1) mad_f = (float)mad;
            uint32_t* row = reinterpret_cast<uint32_t*>(out + r * 36);
            put_f32(row, 0, height);
            put_f32(row, 1, amp);
            put_f32(row, 2, area_f);
            put_f32(row, 3, mad_f);
            put_i64(row, 4, rec.ts[r]);
            row[6] = (uint32_t)(uint16_t)rec.board[r] | ((uint32_t)(uint16_t)rec.chan[r] << 16);
            put_i64(row, 7, r);
        }
    }
}

// =============================================================================================
// K6: integral-quantile width  (waveform_width_integral.py:166-227)
// =============================================================================================
template <int SRC>
__device__ __forceinline__ double width_x(const WaveSrc<SRC>& src, int i, bool known, int pol,
                                          float b32, double baseline) {
    double s;
    if (known) {
        const float d = src.at_f32(i) - b32;
        s = (double)(pol == WFA_POL_POSITIVE ? d : -d);
    } else {
        s = -(src.at(i) - baseline);
    }
    return s > 0.0 ? s : 0.0;
}

template <int SRC>
__global__ __launch_bounds__(kBlock) void k_width_integral(PoolView pool, RecView rec, SgParams sg,
                                                           WidthParams wp, uint8_t* __restrict__ out) {
    const int lane = lane_id();
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t r = wave0; r < rec.R; r += nwaves) {
        const int L = rec.len[r];
        const int64_t off = rec.off[r];
        WaveSrc<SRC> src = make_src<SRC>(pool, sg, off, L);
        const double baseline = rec.baseline[r];
        const float b32 = (float)baseline;
        const int pol = rec.pol[r];
        const bool known = pol == WFA_POL_NEGATIVE || pol == WFA_POL_POSITIVE;

        double q = 0.0;
        for (int i = lane; i < L; i += kWave) q += width_x<SRC>(src, i, known, pol, b32, baseline);
        q = wave_sum(q);

        int lo_i = 0, hi_i = 0;
        const bool ok = q > 0.0 && q <= 1.7976931348623157e308;  // finite and positive
        if (ok) {
            const double t_lo = wp.q_low * q, t_hi = wp.q_high * q;
            lo_i = -1; hi_i = -1;
            double carry = 0.0;
            for (int base = 0; base < L && (lo_i < 0 || hi_i < 0); base += kWave) {
                const int i = base + lane;
                const double x = i < L ? width_x<SRC>(src, i, known, pol, b32, baseline) : 0.0;
                double inc = x;
#pragma unroll
                for (int d = 1; d < kWave; d <<= 1) {
                    const double o = __shfl_up(inc, d, kWave);
                    if (lane >= d) inc += o;
                }
                const double c = carry + inc;
                const bool valid = i < L;
                if (lo_i < 0) {
                    const uint64_t m = __ballot(valid && c >= t_lo);
                    if (m) lo_i = base + __ffsll((long long)m) - 1;
                }
                if (hi_i < 0) {
                    const uint64_t m = __ballot(valid && c >= t_hi);
                    if (m) hi_i = base + __ffsll((long long)m) - 1;
                }
                carry = __shfl(c, kWave - 1, kWave);
            }
            if (lo_i < 0) lo_i = L;  // np.searchsorted returns len(cumsum)
            if (hi_i < 0) hi_i = L;
        }
        if (lane == 0) {
            const double lo = (double)lo_i, hi = (double)hi_i;
            const double w = (double)(hi_i - lo_i > 0 ? hi_i - lo_i : 0);
            uint32_t* row = reinterpret_cast<uint32_t*>(out + r * 52);
            put_f32(row, 0, (float)(lo * wp.dt));
            put_f32(row, 1, (float)(hi * wp.dt));
            put_f32(row, 2, (float)(w * wp.dt));
            put_f32(row, 3, (float)lo);
            put_f32(row, 4, (float)hi);
            put_f32(row, 5, (float)w);
            put_f64(row, 6, q);
            put_i64(row, 8, rec.ts[r]);
            row[10] = (uint32_t)(uint16_t)rec.board[r] | ((uint32_t)(uint16_t)rec.chan[r] << 16);
            put_i64(row, 11, r);
        }
    }
}

// =============================================================================================
// launchers
// =============================================================================================
static inline int grid_for_records(int64_t R) {
    int64_t g = (R + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;  // 256 CUs x 8 blocks of 4 waves
    return (int)g;
}

hipError_t launch_baseline_mean(hipStream_t st, const PoolView& pool, const RecView& rec,
                                int32_t start, int32_t end, double* out) {
    hipLaunchKernelGGL(k_baseline_mean, dim3(grid_for_records(rec.R)), dim3(kBlock), 0, st, pool, rec,
                       start, end, out);
    return hipGetLastError();
}

hipError_t launch_savgol(hipStream_t st, const PoolView& pool, const RecView& rec,
                         const SgParams& sg, float* out) {
    hipLaunchKernelGGL(k_savgol, dim3(grid_for_records(rec.R)), dim3(kBlock), 0, st, pool, rec, sg, out);
    return hipGetLastError();
}

int hits_grid(int64_t R) { return grid_for_records(R); }
int hits_waves(int64_t R) { return grid_for_records(R) * kWavesPerBlock; }

hipError_t launch_hits(hipStream_t st, int source, bool fused_baseline, const PoolView& pool,
                       const RecView& rec, const SgParams& sg, const HitParams& hp) {
    const int grid = grid_for_records(rec.R);
    const size_t lds = (size_t)kWavesPerBlock * hp.bm_words * sizeof(uint64_t);
#define WFA_LAUNCH_HITS(SRC, FB) \
    hipLaunchKernelGGL((k_hits<SRC, FB>), dim3(grid), dim3(kBlock), lds, st, pool, rec, sg, hp)
    if (source == WFA_SRC_RAW) {
        if (fused_baseline) WFA_LAUNCH_HITS(WFA_SRC_RAW, true); else WFA_LAUNCH_HITS(WFA_SRC_RAW, false);
    } else if (source == WFA_SRC_F32) {
        WFA_LAUNCH_HITS(WFA_SRC_F32, false);
    } else {
        if (fused_baseline) WFA_LAUNCH_HITS(WFA_SRC_SG_FUSED, true); else WFA_LAUNCH_HITS(WFA_SRC_SG_FUSED, false);
    }
#undef WFA_LAUNCH_HITS
    return hipGetLastError();
}

int64_t scan_blocks_for(int64_t n) { return (n + kScanTile - 1) / kScanTile; }

hipError_t launch_scan(hipStream_t st, const int32_t* counts, int64_t n, int64_t* block_sums,
                       int64_t* out) {
    const int64_t nb = scan_blocks_for(n);
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(kBlock), 0, st, counts, n, block_sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, st, block_sums, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(kBlock), 0, st, counts, n, block_sums, out);
    return hipGetLastError();
}

hipError_t launch_hits_gather(hipStream_t st, const uint8_t* tmp, const int64_t* tmp_start,
                              const int32_t* nhits, const int64_t* out_start, int64_t R, uint8_t* out) {
    if (R == 0) return hipSuccess;
    const unsigned grid = (unsigned)((R + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_hits_gather, dim3(grid), dim3(kBlock), 0, st, tmp, tmp_start, nhits, out_start, R, out);
    return hipGetLastError();
}

hipError_t launch_basic_features(hipStream_t st, int source, const PoolView& pool, const RecView& rec,
                                 const SgParams& sg, const FeatParams& fp, uint8_t* out) {
    const int grid = grid_for_records(rec.R);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_basic_features<WFA_SRC_RAW>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, fp, out);
    else if (source == WFA_SRC_F32)
        hipLaunchKernelGGL((k_basic_features<WFA_SRC_F32>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, fp, out);
    else
        hipLaunchKernelGGL((k_basic_features<WFA_SRC_SG_FUSED>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, fp, out);
    return hipGetLastError();
}

hipError_t launch_width_integral(hipStream_t st, int source, const PoolView& pool, const RecView& rec,
                                 const SgParams& sg, const WidthParams& wp, uint8_t* out) {
    const int grid = grid_for_records(rec.R);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_width_integral<WFA_SRC_RAW>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, wp, out);
    else if (source == WFA_SRC_F32)
        hipLaunchKernelGGL((k_width_integral<WFA_SRC_F32>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, wp, out);
    else
        hipLaunchKernelGGL((k_width_integral<WFA_SRC_SG_FUSED>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, wp, out);
    return hipGetLastError();
}

}  // namespace wfa
