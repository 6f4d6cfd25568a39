// HIP kernels (gfx950 / CDNA4) for the per-record waveform hot path.
//
// Every kernel works on the device-resident pool + records SoA of a wfa_ctx.  Work unit = one
// record per 64-lane wavefront (records are independent: SURVEY.md section 8e); blocks of 4
// waves grid-stride over the records.  Arithmetic follows the reference literally (float64
// where the reference computes in float64, float32 where it rounds to float32); the file is
// compiled with -ffp-contract=off so no multiply-add is fused that the reference does not fuse.
//
// Reference lines are cited at each device function (paths relative to waveform_analysis/).

#include <type_traits>

#include "wfa_kernels.hpp"
#include "wfa_numpy.hpp"
#include "wfa_device.hpp"

namespace wfa {


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


// =============================================================================================
// K7 fast path: fused baseline + Savitzky-Golay + threshold hits in exact integer arithmetic
// =============================================================================================
// For uint16 samples the SG output is the exact rational  y = (n . x) / den  with integer n
// (sg_plan.py).  The reference thresholds  sig = +-(b - f32(y)) >= thr ; float32 rounding and the
// float64 subtraction are monotone, so the mask is  Z <= Zc  for an integer Zc, up to a band of
// `margin` numerator units that covers the float32 rounding of y.
//
// The pass is split so that the HBM-streaming part stays lean (no LDS, few registers):
//   A  k_sg_mask   one wave per record: 16-byte loads (next record's tiles already in flight),
//                  halo dwords over DPP wave shifts, 8 outputs x (H+1) v_dot2_i32_i16 per lane,
//                  one v_cmp per sample against the candidate bound.  Candidates inside the band
//                  ("borderline", rare) and the 2H edge samples are decided with the reference's
//                  float64 code.  Writes 1 mask bit per sample + the run count per record.
//      scan        exclusive scan of the run counts = row offsets in (record, start) order.
//   B1 k_hit_runs  one lane per record: bit scan of its mask words -> (record, start, end) rows.
//   B2 k_hit_rows  one lane per hit: re-reads the hit window (L2 / Infinity-Cache resident),
//                  y = f32((n . x)/den) exactly as above (float64 chain below the guard),
//                  float64 signal, first-argmax, integral, and the packed 60-byte row.


struct Tile {
    uint32_t d[4];
};

__device__ __forceinline__ Tile load_tile(const uint16_t* __restrict__ pool, int64_t a0, int t, int lane,
                                          int64_t rec_lo, int64_t rec_hi, uint32_t fill) {
    // aligned 16-byte chunk [c, c+8) samples; loaded only if it overlaps the record
    const int64_t c = a0 + (int64_t)t * 512 + lane * 8;
    Tile r;
    if (c + 8 > rec_lo && c < rec_hi) {
        const uint4 v = *reinterpret_cast<const uint4*>(pool + c);
        r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w;
    } else {
        r.d[0] = r.d[1] = r.d[2] = r.d[3] = fill;
    }
    return r;
}

struct RecP {  // wave-uniform record parameters (scalar loads)
    int64_t off;
    int L;
    int pol;
    double thr;
    double bl;
};
__device__ __forceinline__ RecP load_recp(const RecView& rec, int64_t r, bool want_bl) {
    RecP p;
    p.off = rec.off[r];
    p.L = rec.len[r];
    p.pol = rec.pol[r];
    p.thr = rec.thr[r];
    p.bl = want_bl ? rec.baseline[r] : 0.0;
    return p;
}


// 8 SG numerators of one lane's chunk from its own dwords and the halo dwords of both neighbours.
//   E[0..3] = previous chunk, E[4..7] = own, E[8..11] = next  (samples biased by -32768)
template <int W>
__device__ __forceinline__ void sg_chunk_numerators(const uint32_t (&E)[12], const uint32_t* cpm, int (&Z)[8]) {
    constexpr int H = W / 2;
    constexpr int NP = H + 1;
    uint32_t S[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) S[k] = __builtin_amdgcn_alignbit(E[k + 1], E[k], 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ws = j - H;  // first sample of the window, relative to the chunk
        int acc = 0;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const uint32_t pair = (ws & 1) == 0 ? E[ws / 2 + m + 4] : S[(ws - 1) / 2 + m + 4];
            acc = sdot2_acc(pair, cpm[m], acc);
        }
        Z[j] = acc;
    }
}

// The same 8 numerators plus `addend`, first tap alone: a 16 x 16 + 32 multiply-add takes the addend from another
// register (the 2-address v_dot2c needs its accumulator initialised: one v_mov per output), then the taps 1 .. W - 1 as
// H pairs.  c0 = n[0], cq[m] = (n[2m + 1], n[2m + 2]) packed.
template <int W>
__device__ __forceinline__ void sg_chunk_numerators_add(const uint32_t (&E)[12], int c0, const uint32_t* cq, int addend,
                                                        int (&Z)[8]) {
    constexpr int H = W / 2;
    uint32_t S[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) S[k] = __builtin_amdgcn_alignbit(E[k + 1], E[k], 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ws = j - H + 8;  // first sample of the window, counted from E[0]'s first sample (>= 1)
        // sample ws is the low half of E[ws / 2] (ws even) or of S[(ws - 1) / 2] (ws odd); the pairs behind it sit in
        // the array of the other parity
        const uint32_t x0 = (ws & 1) == 0 ? E[ws / 2] : S[(ws - 1) / 2];
        int acc;
        asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(acc) : "v"(x0), "s"(c0), "v"(addend));
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const uint32_t pair = (ws & 1) == 0 ? S[ws / 2 + m] : E[(ws + 1) / 2 + m];
            acc = sdot2_acc(pair, cq[m], acc);
        }
        Z[j] = acc;
    }
}

// PF = tiles of the next record kept in flight per wave (PF x 1 KiB)
template <int W, bool FUSED_BASELINE, int PF>
__global__ __launch_bounds__(kBlock) void k_sg_mask(PoolView pool, RecView rec, SgParams sg, MaskParams mp) {
    constexpr int H = W / 2;
    constexpr int NP = H + 1;
    static_assert(W % 2 == 1 && W >= 3 && W <= 15, "fast path needs the halo inside the adjacent lane");
    const int lane = lane_id();
    const int64_t wave0 = uniform_i64((int64_t)blockIdx.x * kWavesPerBlock + wave_in_block());
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;

    // coefficient pairs (n[2m], n[2m+1]) as packed int16
    uint32_t cp_pos[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        const int n0 = sg.itab[2 * m];
        const int n1 = (2 * m + 1 < W) ? sg.itab[2 * m + 1] : 0;
        cp_pos[m] = ((uint32_t)n0 & 0xffffu) | ((uint32_t)n1 << 16);
    }
    const double bias = 32768.0 * (double)sg.den;  // sum(n) == den for a smoothing filter
    const double dden = (double)sg.den, dden_e = (double)sg.den_edge;
    const int margin = sg.margin, margin_e = sg.margin_edge;

    // integer rows of the edge projection (2H rows x W), one row per edge lane
    __shared__ int32_t etab[2 * H * W];
    for (int k = threadIdx.x; k < 2 * H * W; k += kBlock) etab[k] = sg.itab[W + k];
    __syncthreads();
    int erow[W];
#pragma unroll
    for (int k = 0; k < W; ++k) erow[k] = lane < 2 * H ? etab[lane * W + k] : 0;

    RecP cp{};
    Tile pf[PF];
    if (wave0 < rec.R) {
        cp = load_recp(rec, wave0, !FUSED_BASELINE);
        const int sh0 = (int)(cp.off & 7);
        const uint32_t fill0 = cp.pol == WFA_POL_POSITIVE ? 0u : 0xffffffffu;
#pragma unroll
        for (int t = 0; t < PF; ++t)
            pf[t] = load_tile(pool.u16, cp.off - sh0, t, lane, cp.off, cp.off + cp.L, fill0);
    }

    for (int64_t r = wave0; r < rec.R; r += nwaves) {
        const int64_t rn = r + nwaves;
        const bool has_next = rn < rec.R;
        RecP np{};
        if (has_next) np = load_recp(rec, rn, !FUSED_BASELINE);

        const int L = cp.L;
        const int64_t off = cp.off;
        const bool positive = cp.pol == WFA_POL_POSITIVE;
        const int sh = (int)(off & 7);  // bit index of sample i in the record's mask = i + sh
        const int nbits = L + sh;
        const int nw = (nbits + kWave - 1) / kWave;
        const int T = (nbits + 511) / 512;
        const uint32_t fill_raw = positive ? 0u : 0xffffffffu;  // out-of-record samples: never candidates
        const int64_t a0 = off - sh;
        uint8_t* __restrict__ bm8 = mp.bitmap + rec.bm_off[r];

        WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(pool, sg, off, L);
        double baseline;
        if (FUSED_BASELINE) {
            const int e = mp.bl_end < L ? mp.bl_end : L;
            int64_t s = 0;
            for (int i = mp.bl_start + lane; i < e; i += kWave) s += src.xu[i];
            const int64_t tot = wave_sum_i64(s);  // exact for any window (WFA_MAX_RECORD_SAMPLES * 65535 < 2^53)
            baseline = (e <= mp.bl_start) ? __longlong_as_double(0x7ff8000000000000LL)
                                          : (double)tot / (double)(e - mp.bl_start);
            if (lane == 0) rec.baseline_rw[r] = baseline;
        } else {
            baseline = cp.bl;
        }
        const double thr = cp.thr;
        auto exact_mask = [&](int i) {  // the reference's float64 decision for sample i
            const double w = src.at(i);
            const double sig = positive ? (w - baseline) : (baseline - w);
            return sig >= thr;
        };

        int n_runs = 0;
        if (src.sg.w == W) {
            // ---- edges first (their bits are merged into the lane bytes below) ----------------------
            uint64_t emask = 0;
            {
                bool m = false;
                if (lane < 2 * H) {
                    // integer projection row . first/last W samples (independent loads, one latency)
                    const uint16_t* xe = lane < H ? src.xu : src.xu + (L - W);
                    int xv[W];
#pragma unroll
                    for (int k = 0; k < W; ++k) xv[k] = xe[k];
                    int acc = 0;
#pragma unroll
                    for (int k = 0; k < W; ++k) acc += erow[k] * xv[k];
                    int zhi_e, zlo_e;
                    int_band(positive, baseline, thr, dden_e, 0.0, margin_e, zhi_e, zlo_e);
                    const int ze = positive ? -acc : acc;
                    m = ze < zhi_e;
                    if (m && ze > zlo_e) m = exact_mask(lane < H ? lane : L - 2 * H + lane);  // borderline
                }
                emask = __ballot(m);
            }
            int zhi, zlo;
            int_band(positive, baseline, thr, dden, positive ? bias : -bias, margin, zhi, zlo);
            uint32_t cpm[NP];
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                // negated coefficients for positive polarity: Z = -(n.x - bias)
                const uint32_t lo = (0u - (cp_pos[m] & 0xffffu)) & 0xffffu, hi = (0u - (cp_pos[m] >> 16)) << 16;
                cpm[m] = positive ? (lo | hi) : cp_pos[m];
            }
            const uint32_t fillb = fill_raw ^ 0x80008000u;
            uint32_t p0 = fillb, p1 = fillb, p2 = fillb, p3 = fillb;  // biased dwords of "lane -1"
            uint32_t carry_bit = 0;                                    // mask bit of the sample before the tile
            Tile fillt; fillt.d[0] = fillt.d[1] = fillt.d[2] = fillt.d[3] = fill_raw;
            auto mem_tile = [&](int t) { return load_tile(pool.u16, a0, t, lane, off, off + L, fill_raw); };
            auto do_tile = [&](int t, const Tile& cur, const Tile& nxt) {
                uint32_t E[12];
#pragma unroll
                for (int k = 0; k < 4; ++k) E[4 + k] = cur.d[k] ^ 0x80008000u;
                const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[0], 0) ^ 0x80008000u;
                const uint32_t n1 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[1], 0) ^ 0x80008000u;
                const uint32_t n2 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[2], 0) ^ 0x80008000u;
                const uint32_t n3 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[3], 0) ^ 0x80008000u;
                E[0] = dpp_from_prev_lane(p0, E[4]);
                E[1] = dpp_from_prev_lane(p1, E[5]);
                E[2] = dpp_from_prev_lane(p2, E[6]);
                E[3] = dpp_from_prev_lane(p3, E[7]);
                E[8] = dpp_from_next_lane(n0, E[4]);
                E[9] = dpp_from_next_lane(n1, E[5]);
                E[10] = dpp_from_next_lane(n2, E[6]);
                E[11] = dpp_from_next_lane(n3, E[7]);
                int Z[8];
                sg_chunk_numerators<W>(E, cpm, Z);
                bool cand = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) cand |= Z[j] < zhi;
                uint32_t byte = 0;
                const bool tile_has_edge = emask != 0;  // rare
                if (__ballot(cand) != 0 || tile_has_edge) {
                    const int ib = t * 512 + lane * 8 - sh;  // record index of this lane's sample 0
                    uint32_t border = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool in = (ib + j >= H) && (ib + j < L - H);
                        const bool c = in && (Z[j] < zhi);
                        byte |= (uint32_t)c << j;
                        border |= (uint32_t)(c && Z[j] > zlo) << j;
                    }
                    if (__ballot(border != 0) != 0) {
                        while (border) {  // rare: the reference's float64 arithmetic decides
                            const int j = __ffs((int)border) - 1;
                            border &= border - 1;
                            if (!exact_mask(ib + j)) byte &= ~(1u << j);
                        }
                    }
                    if (tile_has_edge) {
                        uint64_t em = emask;
                        while (em) {
                            const int e = __ffsll((long long)em) - 1;
                            em &= em - 1;
                            const int a = (e < H ? e : L - 2 * H + e) + sh;
                            if ((a >> 9) == t && ((a >> 3) & 63) == lane) byte |= 1u << (a & 7);
                        }
                    }
                    if (__ballot(byte != 0) != 0) {
                        const uint32_t prevb = dpp_from_prev_lane(carry_bit << 7, byte);
                        const uint32_t starts = byte & ~((byte << 1) | (prevb >> 7)) & 0xffu;
                        n_runs += __popc(starts);
                    }
                }
                if (t * 64 + lane < nw * 8) bm8[t * 64 + lane] = (uint8_t)byte;
                carry_bit = ((uint32_t)__builtin_amdgcn_readlane((int)byte, 63) >> 7) & 1u;
                p0 = (uint32_t)__builtin_amdgcn_readlane((int)E[4], 63);
                p1 = (uint32_t)__builtin_amdgcn_readlane((int)E[5], 63);
                p2 = (uint32_t)__builtin_amdgcn_readlane((int)E[6], 63);
                p3 = (uint32_t)__builtin_amdgcn_readlane((int)E[7], 63);
            };
            Tile tail = fillt;  // tile PF when the record is longer than the prefetch
            if (T > PF) tail = mem_tile(PF);
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                if (t < T) {
                    if (t + 1 < PF) do_tile(t, pf[t], t + 1 < T ? pf[t + 1 < PF ? t + 1 : 0] : fillt);
                    else do_tile(t, pf[t], t + 1 < T ? tail : fillt);
                }
            }
            for (int t = PF; t < T; ++t) {
                const Tile cur = tail;
                tail = t + 1 < T ? mem_tile(t + 1) : fillt;
                do_tile(t, cur, tail);
            }
            if (__ballot(n_runs != 0) != 0) n_runs = wave_sum_i32_dpp(n_runs);
        } else {
            // literal phase A (records shorter than the window): one sample per lane
            uint64_t prev_msb = 0;
            for (int wi = 0; wi < nw; ++wi) {
                const int i = wi * kWave + lane - sh;  // bit wi*64+lane <-> sample i
                const bool m = (i >= 0 && i < L) ? exact_mask(i) : false;
                const uint64_t word = __ballot(m);
                if (lane < 8) bm8[wi * 8 + lane] = (uint8_t)(word >> (8 * lane));
                n_runs += __popcll(word & ~((word << 1) | prev_msb));
                prev_msb = word >> 63;
            }
        }
        if (lane == 0) mp.rec_nhits[r] = n_runs;

        // ---- next record: parameters are here, request its first tiles -------------------------------
        if (has_next) {
            const int shn = (int)(np.off & 7);
            const uint32_t filln = np.pol == WFA_POL_POSITIVE ? 0u : 0xffffffffu;
#pragma unroll
            for (int t = 0; t < PF; ++t)
                pf[t] = load_tile(pool.u16, np.off - shn, t, lane, np.off, np.off + np.L, filln);
        }
        cp = np;
    }
}

// ---- A (span mode): uniform-length, contiguous, 16-byte aligned records ----------------------------
// A wave owns a span of up to 64 consecutive records and treats their samples as one stream of
// 512-sample tiles, so every lane does useful work in every tile.  The per-record work (baseline,
// integer bands, the 2H edge samples) is done first with one lane per record; the results sit in
// an LDS table that the tile loop indexes with the lane's record number.
struct SpanTable {  // per wave, in LDS
    int zhi[kWave], zlo[kWave], eb[kWave], nr[kWave];
    double bl[kWave], thr[kWave];
};

// phase 0: lane = record of the span.  Kept out of line so its registers do not add to the tile loop's.
template <int W, bool FUSED_BASELINE, bool PADDED = false>
__device__ __attribute__((noinline)) void span_phase0(const PoolView& pool, const RecView& rec, const SgParams& sg,
                                                      const MaskParams& mp, const int32_t* __restrict__ etab,
                                                      int64_t g_base, int64_t r0, int nrec, int L, bool positive,
                                                      double bias, SpanTable* __restrict__ tab, int S = 0) {
    // PADDED: record stride S (multiple of 16) > L, the last S - L samples of a record's slot are padding; else S = L
    constexpr int H = W / 2;
    if (!PADDED) S = L;
    const int lane = lane_id();
    const int64_t r = r0 + lane;
    int zhi = INT32_MIN, zlo = INT32_MIN, eb = 0;
    double baseline = 0.0, thr = 0.0;
    if (lane < nrec) {
        const uint4* __restrict__ p = reinterpret_cast<const uint4*>(pool.u16) + ((g_base + (int64_t)lane * S) >> 3);
        thr = rec.thr[r];
        if (FUSED_BASELINE) {
            const int s0 = mp.bl_start, e0 = mp.bl_end < L ? mp.bl_end : L;
            int tot = 0;
            for (int c = s0 >> 3; c * 8 < e0; c += 4) {  // 4 independent 16-byte loads per round
                uint4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ((c + k) * 8 < e0) ? p[c + k] : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t d[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int idx = (c + k) * 8 + j;
                        const int x = (int)((d[j >> 1] >> (16 * (j & 1))) & 0xffffu);
                        tot += (idx >= s0 && idx < e0) ? x : 0;
                    }
                }
            }
            baseline = (e0 <= s0) ? __longlong_as_double(0x7ff8000000000000LL) : (double)tot / (double)(e0 - s0);
            rec.baseline_rw[r] = baseline;
        } else {
            baseline = rec.baseline[r];
        }
        int_band(positive, baseline, thr, (double)sg.den, positive ? bias : -bias, sg.margin, zhi, zlo);

        // edges: integer projection rows (LDS, broadcast reads) on the first / last W samples
        int zhi_e, zlo_e;
        int_band(positive, baseline, thr, (double)sg.den_edge, 0.0, sg.margin_edge, zhi_e, zlo_e);
        uint32_t border_e = 0;
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {  // left edge, then right edge (same registers)
            int xw[W];
            if (!PADDED) {  // L % 8 == 0: the first / last 16 samples are two aligned chunks
                const uint4 c0 = side == 0 ? p[0] : p[(L >> 3) - 2];
                const uint4 c1 = side == 0 ? p[1] : p[(L >> 3) - 1];
                const uint32_t dw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const int q0 = k, q1 = 16 - W + k;  // first W / last W of the 16 samples
                    const uint32_t lo = (dw[q0 >> 1] >> (16 * (q0 & 1))) & 0xffffu;
                    const uint32_t hi = (dw[q1 >> 1] >> (16 * (q1 & 1))) & 0xffffu;
                    xw[k] = (int)(side == 0 ? lo : hi);
                }
            } else {  // the first 16 samples, or the last 32 of the slot: the last W samples end S - L before its end
                const int cb = side == 0 ? 0 : (S >> 3) - 4;
                const uint4 c0 = p[cb], c1 = p[cb + 1];
                const uint4 c2 = side == 0 ? c0 : p[cb + 2], c3 = side == 0 ? c1 : p[cb + 3];
                const uint32_t dw[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w,
                                         c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
                const int q_base = side == 0 ? 0 : L - W - cb * 8;
                for (int k = 0; k < W; ++k) {
                    const int qk = q_base + k;
                    uint32_t word = dw[0];
#pragma unroll
                    for (int m = 1; m < 16; ++m) word = (qk >> 1) == m ? dw[m] : word;
                    xw[k] = (int)((word >> (16 * (qk & 1))) & 0xffffu);
                }
            }
#pragma unroll 1
            for (int eh = 0; eh < H; ++eh) {
                const int e = side * H + eh;
                int acc = 0;
#pragma unroll
                for (int k = 0; k < W; ++k) acc += etab[e * W + k] * xw[k];
                const int ze = positive ? -acc : acc;
                const bool m = ze < zhi_e;
                eb |= (int)m << e;
                border_e |= (uint32_t)(m && ze > zlo_e) << e;
            }
        }
        if (border_e) {  // rare: float64 reference code decides
            WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(pool, sg, g_base + (int64_t)lane * S, L);
            while (border_e) {
                const int e = __ffs((int)border_e) - 1;
                border_e &= border_e - 1;
                const double w = src.at(e < H ? e : L - 2 * H + e);
                const double sig = positive ? (w - baseline) : (baseline - w);
                if (!(sig >= thr)) eb &= ~(1 << e);
            }
        }
    }
    tab->zhi[lane] = zhi; tab->zlo[lane] = zlo; tab->eb[lane] = eb; tab->nr[lane] = 0;
    tab->bl[lane] = baseline; tab->thr[lane] = thr;
}

// ---- A (span mode, 16 samples per lane): same kernel with 1024-sample tiles ---------------------------
// The dot2 kernel is VALU-issue bound (~4.2 cycles per wave64 op, tools/valu_rate.hip), and about a third
// of its instructions are per-tile bookkeeping (halo exchange, record tracking, address math, store).
// With two chunks per lane that part is paid once per 1024 samples.  Requires L % 16 == 0.
template <int W>
__device__ __forceinline__ void sg_chunk16_numerators(const uint32_t (&E)[14], const uint32_t* cpm, int (&Z)[16]) {
    // E[0..2] = last 6 samples of the previous lane, E[3..10] = own 16 samples, E[11..13] = next lane's first 6
    constexpr int H = W / 2;
    constexpr int NP = H + 1;
    uint32_t S[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) S[k] = __builtin_amdgcn_alignbit(E[k + 1], E[k], 16);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ws = j - H + 6;  // first sample of the window, counted from E[0]'s first sample (>= 1)
        int acc = 0;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const uint32_t pair = (ws & 1) == 0 ? E[ws / 2 + m < 14 ? ws / 2 + m : 13] : S[(ws - 1) / 2 + m < 13 ? (ws - 1) / 2 + m : 12];
            acc = m == 0 ? sdot2_first(pair, cpm[0]) : sdot2_acc(pair, cpm[m], acc);
        }
        Z[j] = acc;
    }
}

struct Tile16 {
    uint32_t d[8];
};

template <int W, bool FUSED_BASELINE, bool PADDED = false>
__global__ __launch_bounds__(kBlock, WFA_SPAN_WAVES) void k_sg_mask_span16(PoolView pool, RecView rec, SgParams sg,
                                                                            MaskParams mp, SpanParams sp) {
    constexpr int H = W / 2;
    constexpr int NP = H + 1;
    static_assert(W % 2 == 1 && W >= 3 && W <= 11, "halo of 6 samples per side");
    __shared__ SpanTable s_tab[kWavesPerBlock];
    __shared__ int32_t etab[2 * H * W];
    for (int k = threadIdx.x; k < 2 * H * W; k += kBlock) etab[k] = sg.itab[W + k];
    __syncthreads();
    const int lane = lane_id();
    const int wv = wave_in_block();
    SpanTable* tab = &s_tab[wv];
    const int64_t wave0 = uniform_i64((int64_t)blockIdx.x * kWavesPerBlock + wv);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const int L = sp.L;
    // PADDED: record stride S = roundup16(L), the last S - L (< 16) samples of a slot are padding (shadow layout)
    const int S = PADDED ? sp.S : L;
    const int pad = PADDED ? S - L : 0;
    const bool positive = sp.positive != 0;

    uint32_t cpm[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        int n0 = sg.itab[2 * m];
        int n1 = (2 * m + 1 < W) ? sg.itab[2 * m + 1] : 0;
        if (positive) { n0 = -n0; n1 = -n1; }
        cpm[m] = ((uint32_t)n0 & 0xffffu) | ((uint32_t)n1 << 16);
    }
    const uint32_t fill_raw = positive ? 0u : 0xffffffffu;
    const uint32_t fillb = fill_raw ^ 0x80008000u;

    for (int64_t span = wave0; span < sp.n_spans; span += nwaves) {
        const int64_t r0 = span * sp.rs;
        const int nrec = (int)((rec.R - r0) < sp.rs ? (rec.R - r0) : sp.rs);
        const int64_t g_base = sp.off0 + r0 * S;
        span_phase0<W, FUSED_BASELINE, PADDED>(pool, rec, sg, mp, etab, g_base, r0, nrec, L, positive,
                                               32768.0 * (double)sg.den, tab, S);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        const int span_samples = nrec * S;             // multiple of 16
        const int T = (span_samples + 1023) / 1024;
        const uint16_t* __restrict__ span_ptr = pool.u16 + g_base;
        const int last_pos = span_samples - 16;
        auto tile_at = [&](int t) {
            int pos = t * 1024 + lane * 16;
            pos = pos < last_pos ? pos : last_pos;
            const uint4 v0 = *reinterpret_cast<const uint4*>(span_ptr + pos);
            const uint4 v1 = *reinterpret_cast<const uint4*>(span_ptr + pos + 8);
            Tile16 x;
            x.d[0] = v0.x; x.d[1] = v0.y; x.d[2] = v0.z; x.d[3] = v0.w;
            x.d[4] = v1.x; x.d[5] = v1.y; x.d[6] = v1.z; x.d[7] = v1.w;
            return x;
        };
        int rl = (lane * 16) / S;
        int i0 = lane * 16 - rl * S;  // multiple of 16; a lane's 16 samples never straddle record slots (S % 16 == 0)
        uint32_t p5 = fillb, p6 = fillb, p7 = fillb;
        uint32_t carry_msb = 0;
        uint8_t* __restrict__ bm_span = mp.bitmap + sp.bm_off0 + r0 * sp.bm_stride;
        int bm_pos = rl * (int)sp.bm_stride + (i0 >> 3);
        const int bm_wrap = (int)sp.bm_stride - (S >> 3);

        auto do_tile = [&](int t, const Tile16& cur, const Tile16& nxt) {
            const bool in_span = t * 1024 + lane * 16 < span_samples;
            const int rli = in_span ? rl : 0;
            const int zhi = in_span ? tab->zhi[rli] : INT32_MIN;
            uint32_t E[14];
#pragma unroll
            for (int k = 0; k < 8; ++k) E[3 + k] = cur.d[k] ^ 0x80008000u;
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[0], 0) ^ 0x80008000u;
            const uint32_t n1 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[1], 0) ^ 0x80008000u;
            const uint32_t n2 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[2], 0) ^ 0x80008000u;
            E[0] = dpp_from_prev_lane(p5, E[8]);
            E[1] = dpp_from_prev_lane(p6, E[9]);
            E[2] = dpp_from_prev_lane(p7, E[10]);
            E[11] = dpp_from_next_lane(n0, E[3]);
            E[12] = dpp_from_next_lane(n1, E[4]);
            E[13] = dpp_from_next_lane(n2, E[5]);
            int Z[16];
            sg_chunk16_numerators<W>(E, cpm, Z);

            uint64_t cm[16];
            uint64_t any_c = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) { cm[j] = __ballot(Z[j] < zhi); any_c |= cm[j]; }
            const bool first = i0 == 0, last = i0 == S - 16;
            const uint64_t edge_lanes = __ballot(in_span && (first || last));
            uint32_t bits = 0;
            if (any_c != 0 || edge_lanes != 0) {
                // valid bits: not the H edge samples either side (decided in phase 0), not the padding
                const uint32_t vb = (first ? (0xffffu << H) & 0xffffu : 0xffffu) & (last ? 0xffffu >> (H + pad) : 0xffffu);
                const int zlo = tab->zlo[rli];
                // bits from the compare results (the compiler reuses the lane masks of the candidate test);
                // the band test (Z > zlo) stays on lane masks in scalar registers and only becomes lane
                // bits when some lane of the wave is inside the band
                uint64_t any_b = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool c = Z[j] < zhi;
                    bits |= (uint32_t)c << j;
                    any_b |= cm[j] & __ballot(Z[j] > zlo);
                }
                bits &= vb;
                uint32_t border = 0;
                if (any_b != 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) border |= (uint32_t)(Z[j] < zhi && Z[j] > zlo) << j;
                    border &= vb;
                }
                if (any_b != 0 && __ballot(border != 0) != 0) {
                    if (border) {  // rare: the reference's float64 arithmetic decides
                        WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(pool, sg, g_base + (int64_t)rli * S, L);
                        const double baseline = tab->bl[rli], thr = tab->thr[rli];
                        while (border) {
                            const int j = __ffs((int)border) - 1;
                            border &= border - 1;
                            const double w = src.at(i0 + j);
                            const double sig = positive ? (w - baseline) : (baseline - w);
                            if (!(sig >= thr)) bits &= ~(1u << j);
                        }
                    }
                }
                if (in_span && (first || last)) {
                    const uint32_t ebr = (uint32_t)tab->eb[rli];
                    bits |= first ? (ebr & ((1u << H) - 1u)) : ((ebr >> H) << (16 - H - pad)) & 0xffffu;
                }
                if (__ballot(bits != 0) != 0) {
                    uint32_t prevb = dpp_from_prev_lane(carry_msb << 15, bits);
                    if (first) prevb = 0;
                    const uint32_t starts = bits & ~((bits << 1) | (prevb >> 15)) & 0xffffu;
                    if (starts) atomicAdd(&tab->nr[rli], __popc(starts));
                }
            }
            if (in_span) *reinterpret_cast<uint16_t*>(bm_span + bm_pos) = (uint16_t)bits;
            carry_msb = ((uint32_t)__builtin_amdgcn_readlane((int)bits, 63) >> 15) & 1u;
            p5 = (uint32_t)__builtin_amdgcn_readlane((int)E[8], 63);
            p6 = (uint32_t)__builtin_amdgcn_readlane((int)E[9], 63);
            p7 = (uint32_t)__builtin_amdgcn_readlane((int)E[10], 63);
            i0 += 1024;
            bm_pos += 128;
            while (i0 >= S) { i0 -= S; ++rl; bm_pos += bm_wrap; }
        };
        // ring of 3 tiles (2 x 2 KiB in flight per wave), unrolled by 3
        Tile16 ra = tile_at(0), rb = tile_at(1), rc;
        int t = 0;
        for (; t + 3 <= T; t += 3) {
            rc = tile_at(t + 2); do_tile(t, ra, rb);
            ra = tile_at(t + 3); do_tile(t + 1, rb, rc);
            rb = tile_at(t + 4); do_tile(t + 2, rc, ra);
        }
        if (t < T) { rc = tile_at(t + 2); do_tile(t, ra, rb); ++t; }
        if (t < T) { do_tile(t, rb, rc); ++t; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < nrec) mp.rec_nhits[r0 + lane] = tab->nr[lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// ---- K2 (span mode): materialised wave_pool_filtered from exact integer numerators ---------------------
// Same tiling as k_sg_mask_span; every lane turns its 8 numerators into float32 with
// y = f32(f64(n.x) * (1/den)) (equal to scipy's float32 output for |n.x| >= guard) and
// stores 32 contiguous bytes.  Numerators below the guard and the 2H edge samples of every record
// (integer projection rows, literal below their guard) use the float64 code of k_savgol.
// PADDED: input = the padded shadow pool (record stride S = roundup16(L), see wfa_capi: ensure_shadow), output = the
// packed float32 pool (record stride L)
template <int W, bool PADDED = false>
__global__ __launch_bounds__(kBlock) void k_savgol_span(PoolView pool, RecView rec, SgParams sg, SpanParams sp,
                                                        float* __restrict__ out) {
    constexpr int H = W / 2;
    constexpr int NP = H + 1;
    __shared__ float s_edge[kWavesPerBlock][kWave][2 * H];
    __shared__ int32_t etab[2 * H * W];
    // a tile's 512 float32 outputs on their way out: a lane holds 8 consecutive values (32 bytes), and stored as they
    // are every store instruction writes 16 bytes at a 32-byte stride; through this buffer each of the two store
    // instructions writes one contiguous KiB
    __shared__ __attribute__((aligned(16))) float s_out[kWavesPerBlock][512];
    for (int k = threadIdx.x; k < 2 * H * W; k += kBlock) etab[k] = sg.itab[W + k];
    __syncthreads();
    const int lane = lane_id();
    const int wv = wave_in_block();
    const int64_t wave0 = uniform_i64((int64_t)blockIdx.x * kWavesPerBlock + wv);
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const int L = sp.L;
    const int S = PADDED ? sp.S : L;  // input stride
    uint32_t cpm[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        const int n0 = sg.itab[2 * m];
        const int n1 = (2 * m + 1 < W) ? sg.itab[2 * m + 1] : 0;
        cpm[m] = ((uint32_t)n0 & 0xffffu) | ((uint32_t)n1 << 16);
    }
    const int bias_i = 32768 * sg.den;
    const int guard = sg.guard > INT32_MAX ? INT32_MAX : (int)sg.guard;
    const int64_t guard_e = sg.guard_edge;
    const uint32_t fillb = 0x80008000u;

    for (int64_t span = wave0; span < sp.n_spans; span += nwaves) {
        const int64_t r0 = span * sp.rs;
        const int nrec = (int)((rec.R - r0) < sp.rs ? (rec.R - r0) : sp.rs);
        const int64_t g_base = sp.off0 + r0 * S;                   // input
        const int64_t g_out = PADDED ? sp.out_off0 + r0 * L : g_base;  // output
        // ---- edges: lane = record ----
        if (lane < nrec) {
            const uint4* __restrict__ p = reinterpret_cast<const uint4*>(pool.u16) + ((g_base + (int64_t)lane * S) >> 3);
            WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(pool, sg, g_base + (int64_t)lane * S, L);
#pragma unroll 1
            for (int side = 0; side < 2; ++side) {
                int xw[W];
                if (!PADDED) {
                    const uint4 c0 = side == 0 ? p[0] : p[(L >> 3) - 2];
                    const uint4 c1 = side == 0 ? p[1] : p[(L >> 3) - 1];
                    const uint32_t dw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const int q0 = k, q1 = 16 - W + k;
                        const uint32_t lo = (dw[q0 >> 1] >> (16 * (q0 & 1))) & 0xffffu;
                        const uint32_t hi = (dw[q1 >> 1] >> (16 * (q1 & 1))) & 0xffffu;
                        xw[k] = (int)(side == 0 ? lo : hi);
                    }
                } else {  // the first 16 samples, or the last 32 of the slot (the last W samples end S - L before its end)
                    const int cb = side == 0 ? 0 : (S >> 3) - 4;
                    const uint4 c0 = p[cb], c1 = p[cb + 1];
                    const uint4 c2 = side == 0 ? c0 : p[cb + 2], c3 = side == 0 ? c1 : p[cb + 3];
                    const uint32_t dw[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w,
                                             c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
                    const int q_base = side == 0 ? 0 : L - W - cb * 8;
                    for (int k = 0; k < W; ++k) {
                        const int qk = q_base + k;
                        uint32_t word = dw[0];
#pragma unroll
                        for (int m = 1; m < 16; ++m) word = (qk >> 1) == m ? dw[m] : word;
                        xw[k] = (int)((word >> (16 * (qk & 1))) & 0xffffu);
                    }
                }
#pragma unroll 1
                for (int eh = 0; eh < H; ++eh) {
                    const int e = side * H + eh;
                    int acc = 0;
#pragma unroll
                    for (int k = 0; k < W; ++k) acc += etab[e * W + k] * xw[k];
                    float y;
                    if ((int64_t)acc >= guard_e) y = (float)((double)acc * sg.rden_edge);
                    else y = sg_value_f64(src.xu, L, side == 0 ? eh : L - H + eh, src.sg);
                    s_edge[wv][lane][e] = y;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        const int span_samples = nrec * S;
        const int T = (span_samples + 511) / 512;
        const uint16_t* __restrict__ span_ptr = pool.u16 + g_base;
        float* __restrict__ out_span = out + g_out;
        const int last_chunk_pos = span_samples - 8;
        auto tile_at = [&](int t) {
            int pos = t * 512 + lane * 8;
            pos = pos < last_chunk_pos ? pos : last_chunk_pos;
            const uint4 v = *reinterpret_cast<const uint4*>(span_ptr + pos);
            Tile x;
            x.d[0] = v.x; x.d[1] = v.y; x.d[2] = v.z; x.d[3] = v.w;
            return x;
        };
        int rl = (lane * 8) / S;
        int i0 = lane * 8 - rl * S;
        uint32_t p0 = fillb, p1 = fillb, p2 = fillb, p3 = fillb;
        auto do_tile = [&](int t, const Tile& cur, const Tile& nxt) {
            const int pos = t * 512 + lane * 8;
            const bool in_span = pos < span_samples;
            uint32_t E[12];
#pragma unroll
            for (int k = 0; k < 4; ++k) E[4 + k] = cur.d[k] ^ 0x80008000u;
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[0], 0) ^ 0x80008000u;
            const uint32_t n1 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[1], 0) ^ 0x80008000u;
            const uint32_t n2 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[2], 0) ^ 0x80008000u;
            const uint32_t n3 = (uint32_t)__builtin_amdgcn_readlane((int)nxt.d[3], 0) ^ 0x80008000u;
            E[0] = dpp_from_prev_lane(p0, E[4]);
            E[1] = dpp_from_prev_lane(p1, E[5]);
            E[2] = dpp_from_prev_lane(p2, E[6]);
            E[3] = dpp_from_prev_lane(p3, E[7]);
            E[8] = dpp_from_next_lane(n0, E[4]);
            E[9] = dpp_from_next_lane(n1, E[5]);
            E[10] = dpp_from_next_lane(n2, E[6]);
            E[11] = dpp_from_next_lane(n3, E[7]);
            int Z[8];
            sg_chunk_numerators<W>(E, cpm, Z);
            float y[8];
            bool low = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int y_num = Z[j] + bias_i;
                low |= y_num < guard;
                y[j] = (float)((double)y_num * sg.rden);
            }
            // chunks that hold right-edge samples: the last one, or with padding the one or two that overlap [L - H, L)
            const bool first = i0 == 0, last = PADDED ? (i0 + 8 > L - H && i0 < L) : i0 == L - 8;
            if (__ballot(in_span && (low || first || last)) != 0) {
                if (in_span && low) {  // below the integer guard: scipy's float64 chain, literally
                    WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(pool, sg, g_base + (int64_t)rl * S, L);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (Z[j] + bias_i < guard && i0 + j >= H && i0 + j < L - H) y[j] = sg_value_f64(src.xu, L, i0 + j, src.sg);
                }
                if (in_span && first) {
#pragma unroll
                    for (int j = 0; j < H; ++j) y[j] = s_edge[wv][rl][j];
                }
                if (in_span && last) {
                    if (!PADDED) {
#pragma unroll
                        for (int j = 0; j < H; ++j) y[8 - H + j] = s_edge[wv][rl][H + j];
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int e = i0 + j - (L - H);  // index among the H right-edge samples
                            if (e >= 0 && e < H) y[j] = s_edge[wv][rl][H + e];
                        }
                    }
                }
            }
            if (!PADDED) {
                if (t * 512 + 512 <= span_samples) {  // whole tile inside the span (wave-uniform)
                    float4* stage = reinterpret_cast<float4*>(&s_out[wv][0]);
                    stage[2 * lane] = make_float4(y[0], y[1], y[2], y[3]);
                    stage[2 * lane + 1] = make_float4(y[4], y[5], y[6], y[7]);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const float4 a = stage[lane], b = stage[64 + lane];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    f4v* dst = reinterpret_cast<f4v*>(out_span + t * 512);
                    __builtin_nontemporal_store(f4v{a.x, a.y, a.z, a.w}, &dst[lane]);  // written once, read by a later kernel
                    __builtin_nontemporal_store(f4v{b.x, b.y, b.z, b.w}, &dst[64 + lane]);
                } else if (in_span) {
                    float4* dst = reinterpret_cast<float4*>(out_span + pos);
                    dst[0] = make_float4(y[0], y[1], y[2], y[3]);
                    dst[1] = make_float4(y[4], y[5], y[6], y[7]);
                }
            } else if (in_span && i0 < L) {  // packed output: record rl of the span, sample i0 ...
                float* dst = out_span + (int64_t)rl * L + i0;
                if (i0 + 8 <= L && ((g_out + (int64_t)rl * L + i0) & 3) == 0) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f4v{y[0], y[1], y[2], y[3]}, reinterpret_cast<f4v*>(dst));
                    __builtin_nontemporal_store(f4v{y[4], y[5], y[6], y[7]}, reinterpret_cast<f4v*>(dst) + 1);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (i0 + j < L) dst[j] = y[j];
                }
            }
            p0 = (uint32_t)__builtin_amdgcn_readlane((int)E[4], 63);
            p1 = (uint32_t)__builtin_amdgcn_readlane((int)E[5], 63);
            p2 = (uint32_t)__builtin_amdgcn_readlane((int)E[6], 63);
            p3 = (uint32_t)__builtin_amdgcn_readlane((int)E[7], 63);
            i0 += 512;
            while (i0 >= S) { i0 -= S; ++rl; }
        };
        Tile ra = tile_at(0), rb = tile_at(1), rc = tile_at(2), rd;
        int t = 0;
        for (; t + 4 <= T; t += 4) {
            rd = tile_at(t + 3); do_tile(t, ra, rb);
            ra = tile_at(t + 4); do_tile(t + 1, rb, rc);
            rb = tile_at(t + 5); do_tile(t + 2, rc, rd);
            rc = tile_at(t + 6); do_tile(t + 3, rd, ra);
        }
        if (t < T) { rd = tile_at(t + 3); do_tile(t, ra, rb); ++t; }
        if (t < T) { do_tile(t, rb, rc); ++t; }
        if (t < T) { do_tile(t, rc, rd); ++t; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

typedef int wfa_v4i __attribute__((ext_vector_type(4)));


// ---- B1: runs of set bits of every record -> hit descriptors ----------------------------------------
// desc = (record, start, end, flag); flag 0: record filtered with the full SG window (integer row
// kernel), 1: literal float64 row kernel (short record / no integer plan).
#ifndef WFA_RUNS_BLOCK
#define WFA_RUNS_BLOCK 256
#endif
constexpr int kRunsBlock = WFA_RUNS_BLOCK;  // records (= lanes) per block of k_hit_runs
__global__ __launch_bounds__(kRunsBlock) void k_hit_runs(RecView rec, const uint8_t* __restrict__ bitmap,
                                                     const int32_t* __restrict__ nhits,
                                                     const int64_t* __restrict__ out_start,
                                                     int4* __restrict__ desc, RowParams rp) {
    // The mask regions of a block's 256 records are contiguous (bm_off is cumulative): stage them in
    // LDS with coalesced 16-byte loads when they fit, so the per-lane bit scan never waits on HBM.
    // dynamic LDS sized by the launcher to the largest block region (<= 48 KiB): a fixed 48 KiB buffer allowed 3
    // blocks per CU, the 28 KiB an 800-sample run needs allow 5
    extern __shared__ uint4 s_bm[];
    const int kStageBytes = rp.stage_bytes;
    const int64_t r_first = (int64_t)blockIdx.x * kRunsBlock;
    const int64_t r_last = r_first + kRunsBlock < rec.R ? r_first + kRunsBlock : rec.R;  // exclusive
    const int64_t b_first = rec.bm_off[r_first];
    const int64_t b_last = rec.bm_off[r_last - 1] + (((int64_t)rec.len[r_last - 1] + 7 + 63) / 64 * 8 + 8);
    const bool staged = (b_last - b_first) <= kStageBytes;  // b_first is a multiple of 8, regions are 8-byte multiples
    if (staged) {
        const int64_t a_first = b_first & ~15ll;  // 16-byte aligned start (bitmap buffer is 256-byte aligned)
        const int n16 = (int)((b_last - a_first + 15) >> 4);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(bitmap + a_first);
        for (int k = threadIdx.x; k < n16 && k < kStageBytes / 16; k += kRunsBlock) s_bm[k] = src[k];
        __syncthreads();
    }
    const int64_t r = r_first + threadIdx.x;
    if (r >= rec.R) return;
    const int n = nhits[r];
    if (n == 0) return;
    const int L = rec.len[r];
    const int sh = (int)(rec.off[r] & 7);
    const int nbits = L + sh;
    const int nw = (nbits + 63) / 64;
    const bool use_lds = staged && (rec.bm_off[r] - (b_first & ~15ll)) + (int64_t)nw * 8 <= kStageBytes;
    const uint64_t* __restrict__ bm =
        use_lds ? reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint8_t*>(s_bm) + (rec.bm_off[r] - (b_first & ~15ll)))
                : reinterpret_cast<const uint64_t*>(bitmap + rec.bm_off[r]);
    int4* __restrict__ out = desc + out_start[r];
    auto emit = [&](int k, int start, int end) {
        const bool fast = rp.fast_halo > 0 && L >= 2 * rp.fast_halo + 1;  // record filtered with the full window
        if (rp.cap == 0 || out_start[r] + k < rp.cap) out[k] = make_int4((int)r, start, end, fast ? 0 : 1);
    };
    int k = 0;
    int run_start = -1;  // >= 0 while inside a run
    for (int wi = 0; wi < nw && k < n; ++wi) {
        const uint64_t w = bm[wi];
        int pos = 0;  // bits below pos are consumed
        while (pos < 64) {
            const uint64_t rest = (run_start < 0 ? w : ~w) & (~0ull << pos);
            if (rest == 0) break;
            const int b = __ffsll((long long)rest) - 1;
            if (run_start < 0) {
                run_start = wi * 64 + b;
            } else {
                emit(k, run_start - sh, wi * 64 + b - sh);
                ++k;
                run_start = -1;
            }
            pos = b + 1;
        }
    }
    if (run_start >= 0 && k < n) emit(k, run_start - sh, nw * 64 - sh);
}

// ---- B2: hit rows ---------------------------------------------------------------------------------------
// literal path: one lane per hit, float64 code of the reference for every window sample.
// only_flagged: process descriptors with flag != 0 (the fast kernel did the others).
template <int SRC>
__global__ __launch_bounds__(kBlock) void k_hit_rows_literal(PoolView pool, RecView rec, SgParams sg, RowParams rp,
                                                             const int4* __restrict__ desc, int64_t n_hits,
                                                             int only_flagged, uint8_t* __restrict__ out) {
    const int64_t h = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (rp.pass_report && h == 0) {  // (every earlier kernel of the pass has completed: nothing else touches these words now)
        rp.pass_report[0] = rp.n_dev ? *rp.n_dev : n_hits;
        rp.pass_report[1] = (int64_t)rp.pass_ctrl[0];
        rp.pass_report[2] = (int64_t)rp.pass_ctrl[1];
        rp.pass_ctrl[0] = 0ull;
        rp.pass_ctrl[1] = 0ull;
    }
    if (rp.pass_groups)
        for (int64_t q = h; q < rp.pass_n_groups; q += (int64_t)gridDim.x * kBlock) rp.pass_groups[q] = 0ull;
    if (rp.n_dev && *rp.n_dev < n_hits) n_hits = *rp.n_dev;
    int64_t hh = h;
    bool listed = false;
    if (only_flagged && rp.lit_cnt) {
        const uint32_t nl = *rp.lit_cnt;
        if (nl <= (uint32_t)rp.lit_cap) {  // the fast kernel listed every hit it left: no pass over the descriptors' flags
            if (h >= (int64_t)nl) return;
            hh = rp.lit_list[h];
            listed = true;
        }
    }
    if (hh >= n_hits) return;
    const int4 d = desc[hh];
    if (only_flagged && !listed && d.w == 0) return;
    const int64_t r = d.x;
    const int start = d.y, end = d.z;
    const int L = rec.len[r];
    WaveSrc<SRC> src = make_src<SRC>(pool, sg, rec.off[r], L);
    HitCtx hc;
    hc.L = L; hc.max_len = rp.max_len; hc.le = rp.le; hc.re = rp.re;
    hc.thr = rec.thr[r];
    hc.positive = rec.pol[r] == WFA_POL_POSITIVE;
    hc.baseline = rec.baseline[r];
    const int seg_start = start - hc.le > 0 ? start - hc.le : 0;
    const int seg_end = end + hc.re < hc.max_len ? end + hc.re : hc.max_len;
    HitAcc acc{-__builtin_huge_val(), 0x7fffffff, 0.0};
    for (int i = seg_start; i < seg_end; ++i) acc.add(hit_signal<SRC>(src, hc, i), i);
    write_hit_row(out, hh, rec, r, L, start, end, seg_start, seg_end, acc.best_i, acc.best, acc.sum);
}

// fast path: 8 lanes per hit.  Lane q of a group loads the aligned 16-byte chunk (c + q) and produces its 8 outputs
// from exact integer numerators (halo over DPP; the chunks either side of the round from a second load that lanes 0
// and 7 use).  64 window samples per round, one coalesced 128-byte read per group (+ the two halo chunks).
// Hit windows are strongly bimodal (fragments of a few samples vs. pulses of 100-300), and a wave iterates
// as long as its longest hit.  Each 1024-thread block therefore ranks its 128 hits by window length in
// LDS first, so that the 8 hits sharing a wave need about the same number of rounds.
constexpr int kRowsBlock = 256;  // 32 hits ranked per block; 1024 / 512 / 128 / 64 threads measured 0.58 / 0.50 / 0.49 / 0.59 ms against 0.47
constexpr int kRowsHits = kRowsBlock / 8;

template <int W>
__global__ __launch_bounds__(kRowsBlock) void k_hit_rows_grp(PoolView pool, RecView rec, SgParams sg, RowParams rp,
                                                             int4* __restrict__ desc, int64_t n_hits,
                                                             uint8_t* __restrict__ out) {
    constexpr int H = W / 2;
    __shared__ int s_len[kRowsHits];
    __shared__ int s_perm[kRowsHits];
    const int q = threadIdx.x & 7;
    const int grp = threadIdx.x >> 3;
    const int64_t h_base = (int64_t)blockIdx.x * kRowsHits;
    if (rp.n_dev && *rp.n_dev < n_hits) n_hits = *rp.n_dev;
    if (h_base >= n_hits) return;  // whole block beyond the rows of this pass (uniform: before any barrier)
    {
        // window length of the hit in this group's slot (0 = nothing to do), then rank = position in
        // descending order (ties by slot), computed by the group's 8 lanes over 16 slots each
        const int64_t h0 = h_base + grp;
        int len = 0;
        if (h0 < n_hits) {
            const int4 d0 = desc[h0];
            if (d0.w == 0) len = d0.z - d0.y + rp.le + rp.re;
        }
        if (q == 0) s_len[grp] = len;
        __syncthreads();
        int cnt = 0;
        for (int j = q; j < kRowsHits; j += 8) {
            const int lj = s_len[j];
            cnt += (lj > len || (lj == len && j < grp)) ? 1 : 0;
        }
        cnt += dpp_i32(cnt, 0);
        cnt += dpp_i32(cnt, 1);
        cnt += dpp_i32(cnt, 2);
        if (q == 0) s_perm[cnt] = grp;
        __syncthreads();
    }
    const int64_t h = h_base + s_perm[grp];
    const bool live = h < n_hits;
    int4 d = make_int4(0, 0, 0, 1);
    if (live) d = desc[h];
    const bool work = live && d.w == 0;

    const int c0 = sg.itab[0];
    uint32_t cq[H];
#pragma unroll
    for (int m = 0; m < H; ++m) cq[m] = ((uint32_t)sg.itab[2 * m + 1] & 0xffffu) | ((uint32_t)sg.itab[2 * m + 2] << 16);
    const int bias_i = 32768 * sg.den;
    const int guard = sg.guard > INT32_MAX ? INT32_MAX : (int)sg.guard;

    const int64_t r = d.x;
    const int start = d.y, end = d.z;
    int L = 0;
    int64_t off = 0;
    double baseline = 0.0;
    bool positive = false;
    if (work) {
        baseline = rec.baseline[r];
        if (rp.uni_L > 0) {  // nothing else to wait for before the first chunk load
            L = rp.uni_L;
            off = rp.uni_off0 + r * (int64_t)(rp.uni_S ? rp.uni_S : rp.uni_L);
            positive = rp.uni_positive != 0;
        } else {
            L = rec.len[r];
            off = rec.off[r];
            positive = rec.pol[r] == WFA_POL_POSITIVE;
        }
    }
    const int seg_start = start - rp.le > 0 ? start - rp.le : 0;
    const int seg_end = end + rp.re < rp.max_len ? end + rp.re : rp.max_len;
    // interior part of the window (integer numerators); the <= 2H edge samples and the zero padding
    // beyond the record (reference's dense matrix) are evaluated literally below
    const int ilo = seg_start > H ? seg_start : H;
    const int ihi = seg_end < L - H ? seg_end : L - H;
    const bool has_int = work && ihi > ilo;
    const int64_t g0 = off + ilo, g1 = off + ihi;
    const int64_t c_first = g0 >> 3, c_last = has_int ? ((g1 - 1) >> 3) : -1;
    const uint4* __restrict__ p16 = reinterpret_cast<const uint4*>(pool.u16);

    HitAccAny acc{-__builtin_huge_val(), 0x7fffffff, 0.0};
    bool need_literal = false;
    // Interior samples: sig = +-(b - f64(y32)) is strictly monotone in y32 as long as the float64
    // subtraction is exact, which holds for |b| < 2^18 and |y| < 2^17 (difference needs < 53 bits).
    // Then the first maximum of sig is the first extremum of y32: a float32 compare per sample
    // instead of a float64 compare chain; sig itself is only needed for the integral.  Both polarities run the
    // same code on t = +-y32 (one xor on the sign bit): sig = sb - f64(t) with sb = +-b, first minimum of t.
    // Per sample the body is selects only (an earlier form with per-lane branches cost 15 scalar instructions and two
    // jumps per sample); what remains is one wave-level skip per sample index.  A baseline outside the exact range sends
    // the hit to the literal kernel.
    const bool y_order = fabs(baseline) < 262144.0;
    if (work && !y_order) need_literal = true;
    const uint32_t sign_mask = positive ? 0x80000000u : 0u;
    const double sb = positive ? -baseline : baseline;
    float ext_t = __builtin_huge_valf();
    int ext_i = 0x7fffffff;
    const int wlen = ihi - ilo;
    int y_num_min = INT32_MAX;
    // rounds: the whole wave iterates while any group has chunks left
    // unconditional load at a clamped index (a load behind a branch drags an `s_waitcnt vmcnt(0)` with it): chunks
    // outside [0, c_last + 1] only feed samples that are not in the window (record edges, finished groups); the pool
    // buffer has 256 B of slack behind its end
    const int64_t c_hi_load = has_int ? c_last + 1 : 0;
    // all 8 lanes of a group evaluate a chunk (64 window samples per round): the halo in front of the round's first
    // chunk and behind its last one comes from a second load that only lanes 0 and 7 need (the others read their own
    // chunk again, an L1 hit).  With lanes 0 and 7 as pure halo providers a round covered 48 samples for the same
    // per-round cost.
    auto clampc = [&](int64_t m) { return m < 0 ? (int64_t)0 : (m > c_hi_load ? c_hi_load : m); };
    auto load_own = [&](int64_t c) { return p16[clampc(c + q)]; };
    auto load_edge = [&](int64_t c) { return p16[clampc(q == 0 ? c - 1 : (q == 7 ? c + 8 : c + q))]; };
    auto do_round = [&](int64_t c, const uint4& v, const uint4& ve) {
        const int64_t mine = c + q;
        uint32_t E[12];
        E[4] = v.x ^ 0x80008000u; E[5] = v.y ^ 0x80008000u; E[6] = v.z ^ 0x80008000u; E[7] = v.w ^ 0x80008000u;
        const uint32_t X[4] = {ve.x ^ 0x80008000u, ve.y ^ 0x80008000u, ve.z ^ 0x80008000u, ve.w ^ 0x80008000u};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t pv = dpp_from_prev_lane(0u, E[4 + k]), nv = dpp_from_next_lane(0u, E[4 + k]);
            E[k] = q == 0 ? X[k] : pv;
            E[8 + k] = q == 7 ? X[k] : nv;
        }
        int Z[8];  // numerators of the unbiased samples: n . x
        sg_chunk_numerators_add<W>(E, c0, cq, bias_i, Z);
        const bool lane_ok = mine <= c_last;
        const int rel0 = (int)(mine * 8 - g0);  // window-relative index of this chunk's sample 0
        // a sample outside the window gets t = +inf: it never wins the extremum and its signal is -inf, clamped to 0
        const int idx0 = ilo + rel0;
        {
            // integer guard: the smallest numerator of every chunk a working lane evaluates (the samples of a chunk that
            // lie outside the window are ordinary neighbours of the same record: at worst a hit goes to the literal
            // kernel that did not have to)
            int zm = Z[0] < Z[1] ? Z[0] : Z[1];
#pragma unroll
            for (int j = 2; j < 8; ++j) zm = Z[j] < zm ? Z[j] : zm;
            zm = lane_ok ? zm : INT32_MAX;
            y_num_min = zm < y_num_min ? zm : y_num_min;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in = lane_ok && (unsigned)(rel0 + j) < (unsigned)wlen;
            const int y_num = Z[j];
            const float y32 = (float)((double)y_num * sg.rden);
            // (the compiler branches around this block when no lane of the wave has sample j in its window; forcing a
            // straight-line loop measured 0.475 ms against 0.456: short hits leave most of a round's slots empty)
            const float t = in ? __uint_as_float(__float_as_uint(y32) ^ sign_mask) : __builtin_huge_valf();
            const bool better = t < ext_t;  // ascending index: the first extremum is kept
            ext_t = better ? t : ext_t;
            ext_i = better ? idx0 + j : ext_i;
            acc.sum += fmax(sb - (double)t, 0.0);
        }
    };
    // one round's loads in flight ahead of the round being evaluated; a ring of 3 (two ahead) measured 0.480 ms against
    // 0.472: the kernel is bound by instruction issue, not by the latency of its loads
    {
        uint4 v = load_own(c_first), ve = load_edge(c_first);
        for (int64_t c = c_first; __ballot(c <= c_last) != 0; c += 8) {
            const uint4 cur = v, cur_e = ve;
            // the scheduler must not lift the next loads above the reads of `v`: the compiler cannot count a load that
            // is in flight across the back-edge and would wait for `vmcnt(0)`, i.e. for the prefetch it has just issued
            __builtin_amdgcn_sched_barrier(0);
            v = load_own(c + 8);
            ve = load_edge(c + 8);
            __builtin_amdgcn_sched_barrier(0);
            do_round(c, cur, cur_e);
        }
    }
    need_literal |= y_num_min < guard;
    if (ext_i != 0x7fffffff) {
        acc.best = sb - (double)ext_t;
        acc.best_i = ext_i;
    }
    if (__ballot(work && (seg_start < H || seg_end > L - H)) != 0) {
        if (work && (seg_start < H || seg_end > L - H)) {
            WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(pool, sg, off, L);
            HitCtx hc;
            hc.L = L; hc.max_len = rp.max_len; hc.le = rp.le; hc.re = rp.re;
            hc.thr = 0.0; hc.positive = positive; hc.baseline = baseline;
            const int l_end = seg_end < H ? seg_end : H;              // left edge samples [seg_start, l_end)
            for (int i = seg_start + q; i < l_end; i += 8) acc.add(hit_signal<WFA_SRC_SG_FUSED>(src, hc, i), i);
            const int r_beg = seg_start > L - H ? seg_start : L - H;  // right edge + padding [r_beg, seg_end)
            for (int i = r_beg + q; i < seg_end; i += 8) acc.add(hit_signal<WFA_SRC_SG_FUSED>(src, hc, i), i);
        }
    }
    // combine the 8 lanes of the group (butterfly over xor 1, xor 2, mirror)
#pragma unroll
    for (int step = 0; step < 3; ++step) {
        const double ov = dpp_f64(acc.best, step);
        const int oi = dpp_i32(acc.best_i, step);
        const double os = dpp_f64(acc.sum, step);
        const int on = dpp_i32((int)need_literal, step);
        const bool take = (ov > acc.best) || (ov == acc.best && oi < acc.best_i);
        acc.best = take ? ov : acc.best;
        acc.best_i = take ? oi : acc.best_i;
        acc.sum += os;
        need_literal |= on != 0;
    }
    if (work && q == 0) {
        if (need_literal) desc[h].w = 2;  // below the integer guard: the literal kernel redoes this hit
        else write_hit_row(out, h, rec, r, L, start, end, seg_start, seg_end, acc.best_i, acc.best, acc.sum);
    }
}

// fast path, flat form (round 3): the window samples of 64 consecutive hits as ONE list of aligned 8-sample chunks, a chunk per
// lane.  The grouped kernel above gives every hit 8 lanes and iterates as long as the longest hit of a wave: half of all
// hits are fragments of a few samples that fill one or two of their 8 lanes, and it issued 4.4 x the instructions the
// window samples need (profiles/r02_pmc_sq_counters.txt: 257.6e6 for 2.1e6 hits).  Here every wave works on its own 64
// hits, one wave per workgroup, no block barriers:
//   A  lane = hit: window, first chunk, chunk count; a prefix sum of the counts numbers the wave's chunks, a bit per chunk
//      that starts a hit + a popcount turn a chunk number back into its hit;
//   B  lane = chunk, the next round's loads in flight: three 16-byte loads (own chunk + both neighbours), 8 exact
//      numerators, then per hit in LDS with INTEGER atomics (any order gives the same bits)
//        min of (float32 y in unsigned order, sample index)   = the reference's first maximum of the signal,
//        sum of y * 2^23 and count over the samples with signal > 0 (y = float32: an exact integer above the guard),
//      so that  integral = count * (+-baseline) - sum(y);
//   C  lane = hit: the <= 2H edge samples / zero padding of the hits that have any go through a second, short chunk-style
//      list (one sample per lane, literal float64 code), then the row.
// Measured on the way (2.12e6 hits): 64 hits per 256-thread block with phases between __syncthreads 0.60 ms (74 % of the
// wave cycles waiting); 16 hits per wave 0.46; + next round's loads in flight 0.42; 64 hits per wave with per-chunk pairs
// in LDS walked by lane = hit 0.39 (54 instructions per hit, 30 of them the serial walks and one-lane edge loops).
constexpr int kFlatHits = 64;    // per wave
constexpr int kFlatCap = 2048;   // chunks per batch and wave (a start bit each)

struct FlatHit {  // 48 bytes: three 16-byte LDS reads per chunk
    int64_t cfirst;   // first chunk of the hit's interior window (pool index / 8)
    double sb;        // +-baseline
    int g0lo;         // first window sample inside that chunk (0..7)
    int wlen;         // interior window length
    int ilo;          // record index of the first interior window sample
    uint32_t sign;    // sign bit applied to y (positive polarity: first minimum of -y)
    int P;            // chunks of the wave in front of this hit
    float tb;         // largest float32 below sb: signal > 0  <=>  t <= tb
    int L;            // record length (edge list)
    int seg_start;    // window start (edge list)
};
struct FlatLds {  // per wave
    FlatHit hit[kFlatHits];
    unsigned long long key[kFlatHits];   // min over the hit's samples of (ordered float32 t, sample index)
    unsigned long long tsum[kFlatHits];  // sum of t * 2^23 over the samples with signal > 0 (two's complement)
    uint32_t npos[kFlatHits];            // how many
    uint32_t flag[kFlatHits];            // 1: a numerator below the integer guard -> literal kernel
    uint32_t start_bits[kFlatCap / 32];  // chunk f of the batch starts (continues, for bit 0) a hit
    uint32_t word_rank[kFlatCap / 32];   // start bits in front of that word
    uint8_t by_rank[kFlatHits];          // k-th hit of the batch that has chunks
    double esig[kWave];                  // edge list: signal of one sample per lane
};

// chunk (or edge sample) number f of the current batch -> hit
__device__ __forceinline__ int flat_hit_of(const FlatLds* lds, int fb) {
    const uint32_t w = lds->start_bits[fb >> 5];
    const int rank = (int)lds->word_rank[fb >> 5] + __popc(w & (0xffffffffu >> (31 - (fb & 31)))) - 1;
    return lds->by_rank[rank];
}
// start bits of a batch: every lane offers the stretch [lo, hi) of its hit inside the batch window [f0, f0 + cap)
__device__ __forceinline__ void flat_index_batch(FlatLds* lds, int lane, int f0, int lo, int hi) {
    constexpr int kWords = kFlatCap / 32;
    static_assert(kWords == kWave, "one start word per lane");
    lds->start_bits[lane] = 0u;
    const bool has = hi > lo;
    const unsigned long long m = __ballot(has);
    if (has) lds->by_rank[__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (has) atomicOr(&lds->start_bits[(lo - f0) >> 5], 1u << ((lo - f0) & 31));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int total;
    lds->word_rank[lane] = (uint32_t)wave_excl_scan_i32(__popc(lds->start_bits[lane]), total);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int W>
__global__ __launch_bounds__(kWave) void k_hit_rows_flat(PoolView pool, RecView rec, SgParams sg, RowParams rp,
                                                          int4* __restrict__ desc, int64_t n_hits,
                                                          uint8_t* __restrict__ out) {
    constexpr int H = W / 2;
    __shared__ __attribute__((aligned(16))) FlatLds s_lds;  // one wave per workgroup: its LDS and registers are free the moment it ends
    FlatLds* __restrict__ lds = &s_lds;
    const int lane = lane_id();
    const int64_t h_base = (int64_t)blockIdx.x * kFlatHits;
    if (rp.n_dev && *rp.n_dev < n_hits) n_hits = *rp.n_dev;
    if (h_base >= n_hits) return;  // whole wave beyond the rows of this pass

    const int c0 = sg.itab[0];
    uint32_t cq[H];
#pragma unroll
    for (int m = 0; m < H; ++m) cq[m] = ((uint32_t)sg.itab[2 * m + 1] & 0xffffu) | ((uint32_t)sg.itab[2 * m + 2] << 16);
    const int bias_i = 32768 * sg.den;
    const int guard = sg.guard > INT32_MAX ? INT32_MAX : (int)sg.guard;

    // ---- A: lane = hit ----
    const int64_t h = h_base + lane;
    int4 d = make_int4(0, 0, 0, 1);
    if (h < n_hits) d = desc[h];
    const bool work = h < n_hits && d.w == 0;
    const int64_t r = d.x;
    const int start = d.y, end = d.z;
    int L = 0;
    int64_t off = 0;
    double baseline = 0.0;
    bool positive = false;
    if (work) {
        baseline = rec.baseline[r];
        if (rp.uni_L > 0) {
            L = rp.uni_L;
            off = rp.uni_off0 + r * (int64_t)(rp.uni_S ? rp.uni_S : rp.uni_L);
            positive = rp.uni_positive != 0;
        } else {
            L = rec.len[r];
            off = rec.off[r];
            positive = rec.pol[r] == WFA_POL_POSITIVE;
        }
    }
    const int seg_start = start - rp.le > 0 ? start - rp.le : 0;
    const int seg_end = end + rp.re < rp.max_len ? end + rp.re : rp.max_len;
    // interior part of the window (integer numerators); the <= 2H edge samples and the zero padding beyond the record
    // (reference's dense matrix) are evaluated literally in C
    const int ilo = seg_start > H ? seg_start : H;
    const int ihi = seg_end < L - H ? seg_end : L - H;
    // sig = +-(b - f64(y32)) is strictly monotone in y32 while the float64 subtraction is exact (|b| < 2^18, |y| < 2^17):
    // the first maximum of sig is then the first extremum of y32 in float32 order; other baselines -> literal kernel
    const bool y_order = fabs(baseline) < 262144.0;
    bool need_literal = work && !y_order;
    const bool has_int = work && y_order && ihi > ilo;
    const int64_t g0 = off + ilo, g1 = off + ihi;
    const int64_t c_first = g0 >> 3, c_last = has_int ? ((g1 - 1) >> 3) : -1;
    const int n_chunks = has_int ? (int)(c_last - c_first + 1) : 0;
    const double sb = positive ? -baseline : baseline;
    int C;
    const int my_P = wave_excl_scan_i32(n_chunks, C);
    {
        // signal of a sample > 0  <=>  (double)t < sb  <=>  t <= tb, the largest float32 below sb
        float tb = (float)sb;
        if ((double)tb >= sb) {
            const uint32_t u = __float_as_uint(tb);
            tb = __uint_as_float(tb > 0.0f ? u - 1u : (tb < 0.0f ? u + 1u : 0x80000001u));
        }
        FlatHit fh;
        fh.cfirst = c_first; fh.sb = sb; fh.g0lo = (int)(g0 - (c_first << 3)); fh.wlen = ihi - ilo; fh.ilo = ilo;
        fh.sign = positive ? 0x80000000u : 0u; fh.P = my_P; fh.tb = tb; fh.L = L; fh.seg_start = seg_start;
        lds->hit[lane] = fh;
        lds->key[lane] = ~0ull;
        lds->tsum[lane] = 0ull;
        lds->npos[lane] = 0u;
        lds->flag[lane] = 0u;
    }
    const uint4* __restrict__ p16 = reinterpret_cast<const uint4*>(pool.u16);

    for (int f0 = 0; f0 < C; f0 += kFlatCap) {
        const int f1 = f0 + kFlatCap < C ? f0 + kFlatCap : C;
        flat_index_batch(lds, lane, f0, my_P > f0 ? my_P : f0, my_P + n_chunks < f1 ? my_P + n_chunks : f1);
        // ---- B: lane = chunk; the loads of the next round of 64 chunks are in flight while this one is evaluated ----
        // (unconditional loads at a clamped chunk number: a load behind a branch drags an `s_waitcnt vmcnt(0)` with it)
        struct Fetched { FlatHit fh; uint4 vp, v, vn; int j, hh; };
        auto fetch = [&](int f) {
            Fetched x;
            const int fc = f < f1 ? f : f1 - 1;
            x.hh = flat_hit_of(lds, fc - f0);
            x.fh = lds->hit[x.hh];
            x.j = fc - x.fh.P;
            const int64_t c = x.fh.cfirst + x.j;
            x.vp = p16[c > 0 ? c - 1 : 0]; x.v = p16[c]; x.vn = p16[c + 1];  // neighbours: same record (interior window)
            return x;
        };
        Fetched nxt = fetch(f0 + lane);
        for (int f = f0 + lane; __ballot(f < f1) != 0; f += kWave) {
            const Fetched cur = nxt;
            __builtin_amdgcn_sched_barrier(0);
            nxt = fetch(f + kWave);
            __builtin_amdgcn_sched_barrier(0);
            const FlatHit& fh = cur.fh;
            const uint4 vp = cur.vp, v = cur.v, vn = cur.vn;
            uint32_t E[12];
            E[0] = vp.x ^ 0x80008000u; E[1] = vp.y ^ 0x80008000u; E[2] = vp.z ^ 0x80008000u; E[3] = vp.w ^ 0x80008000u;
            E[4] = v.x ^ 0x80008000u; E[5] = v.y ^ 0x80008000u; E[6] = v.z ^ 0x80008000u; E[7] = v.w ^ 0x80008000u;
            E[8] = vn.x ^ 0x80008000u; E[9] = vn.y ^ 0x80008000u; E[10] = vn.z ^ 0x80008000u; E[11] = vn.w ^ 0x80008000u;
            int Z[8];  // numerators of the unbiased samples: n . x
            sg_chunk_numerators_add<W>(E, c0, cq, bias_i, Z);
            // integer guard: the smallest numerator of the chunk (samples of it outside the window are ordinary neighbours of
            // the same record: at worst a hit goes to the literal kernel that did not have to).  Above it |y| >= 1, so every
            // float32 y is a multiple of 2^-23.
            int zm = Z[0] < Z[1] ? Z[0] : Z[1];
#pragma unroll
            for (int k = 2; k < 8; ++k) zm = Z[k] < zm ? Z[k] : zm;
            const int rel0 = 8 * cur.j - fh.g0lo;  // window-relative index of this chunk's sample 0
            float ext_t = __builtin_huge_valf();
            int ext_k = 0;
            double tsum = 0.0;
            int npos = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool in = (unsigned)(rel0 + k) < (unsigned)fh.wlen;
                const float y32 = (float)((double)Z[k] * sg.rden);
                // a sample outside the window gets t = +inf: it never wins the extremum and its signal is not > 0
                const float t = in ? __uint_as_float(__float_as_uint(y32) ^ fh.sign) : __builtin_huge_valf();
                const bool better = t < ext_t;  // ascending index: the first extremum is kept
                ext_t = better ? t : ext_t;
                ext_k = better ? k : ext_k;
                const bool pos = t <= fh.tb;
                tsum += (double)(pos ? t : 0.0f);  // exact: 8 float32 values
                npos += pos ? 1 : 0;
            }
            if (f < f1) {
                if (zm < guard) atomicOr(&lds->flag[cur.hh], 1u);
                // float32 order as unsigned order: negative values with all bits flipped, the others with the sign bit set
                const uint32_t u = __float_as_uint(ext_t);
                const uint32_t ord = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
                atomicMin(&lds->key[cur.hh], ((unsigned long long)ord << 32) | (uint32_t)(fh.ilo + rel0 + ext_k));
                // tsum * 2^23 as a 64-bit integer (|tsum| < 2^20: high part < 2^22, low part < 2^21)
                const double x = tsum * 8388608.0;
                const int hi = (int)(x * (1.0 / 2097152.0));
                const int lo = (int)(x - (double)hi * 2097152.0);
                atomicAdd(&lds->tsum[cur.hh], (unsigned long long)((long long)hi * 2097152ll + (long long)lo));
                atomicAdd(&lds->npos[cur.hh], (uint32_t)npos);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // ---- C: lane = hit ----
    need_literal |= work && lds->flag[lane] != 0u;
    HitAccAny acc{-__builtin_huge_val(), 0x7fffffff, 0.0};
    {
        const unsigned long long key = lds->key[lane];
        if (work && key != ~0ull) {
            const uint32_t ord = (uint32_t)(key >> 32);
            const uint32_t u = ord ^ ((ord >> 31) ? 0x80000000u : 0xffffffffu);
            acc.best = sb - (double)__uint_as_float(u);
            acc.best_i = (int)(uint32_t)key;
            acc.sum = (double)lds->npos[lane] * sb - (double)(long long)lds->tsum[lane] * (1.0 / 8388608.0);
        }
    }
    // edge list: the samples [seg_start, min(seg_end, H)) and [max(seg_start, L - H), seg_end) of every hit that has any,
    // one per lane, evaluated with the reference's float64 code; their hits then take them in ascending order
    {
        const bool edgy = work && !need_literal;
        const int l_end = seg_end < H ? seg_end : H;
        const int nl = edgy && l_end > seg_start ? l_end - seg_start : 0;        // left edge samples
        const int r_beg = seg_start > L - H ? seg_start : L - H;
        const int nr = edgy && seg_end > r_beg ? seg_end - r_beg : 0;            // right edge samples + padding
        int E_tot;
        const int my_E = wave_excl_scan_i32(nl + nr, E_tot);
        for (int e0 = 0; e0 < E_tot; e0 += kWave) {
            const int e1 = e0 + kWave < E_tot ? e0 + kWave : E_tot;
            const int lo = my_E > e0 ? my_E : e0, hi = my_E + nl + nr < e1 ? my_E + nl + nr : e1;
            flat_index_batch(lds, lane, e0, lo, hi);
            // the edge list's prefix of a hit is carried by the hit's lane; ds_bpermute brings it to the sample's lane (all
            // lanes take part: an inactive source lane would deliver nothing)
            const int hh = flat_hit_of(lds, e0 + lane < e1 ? lane : e1 - e0 - 1);
            const int my_E_h = __shfl(my_E, hh, kWave), nl_h = __shfl(nl, hh, kWave), r_beg_h = __shfl(r_beg, hh, kWave);
            if (e0 + lane < e1) {
                const FlatHit fh = lds->hit[hh];
                const int k = e0 + lane - my_E_h;
                const int i = k < nl_h ? fh.seg_start + k : r_beg_h + (k - nl_h);
                const int64_t off_h = (fh.cfirst << 3) + fh.g0lo - fh.ilo;
                WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(pool, sg, off_h, fh.L);
                HitCtx hc;
                hc.L = fh.L; hc.max_len = rp.max_len; hc.le = rp.le; hc.re = rp.re;
                hc.thr = 0.0; hc.positive = fh.sign != 0u; hc.baseline = fh.sign ? -fh.sb : fh.sb;
                lds->esig[lane] = hit_signal<WFA_SRC_SG_FUSED>(src, hc, i);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int e = lo; e < hi; ++e) {
                const int k = e - my_E;
                acc.add(lds->esig[e - e0], k < nl ? seg_start + k : r_beg + (k - nl));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    if (work) {
        if (need_literal) {  // below the integer guard: the literal kernel redoes this hit
            desc[h].w = 2;
            if (rp.lit_cnt) {
                const uint32_t slot = atomicAdd(rp.lit_cnt, 1u);
                if (slot < (uint32_t)rp.lit_cap) rp.lit_list[slot] = (int32_t)h;
            }
        } else {
            write_hit_row(out, h, rec, r, L, start, end, seg_start, seg_end, acc.best_i, acc.best, acc.sum);
        }
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
// K5 / K6: per-record features, one lane per record, sequential semantics
// =============================================================================================
// The reference computes these with numpy reductions whose float64 rounding depends on the order of
// the additions.  One lane walks its record in sample order and performs literally the same sequence:
//   np.sum      = sum over 8192-element blocks of numpy's pairwise sum (128-element leaves with 8
//                 interleaved accumulators, halves split at multiples of 8)         -> `Pairwise`
//   np.cumsum   = strictly sequential                                               -> running sum
//   np.searchsorted(side="left") = first index with cumsum >= target
// so area / q_total / the quantile indices are bit-identical, not merely within tolerance.
// Loads are 16-byte chunks per lane (records of neighbouring lanes are apart, lines are reused from
// L1/L2 over the next chunks).

constexpr int kFeatBlock = 128;  // threads per block (LDS stacks are per thread)

struct PairwiseStacks {
    // numpy's 8192-element reduce block halves down to <= 128 in 6 splits: at most 2 pending entries per level + 1
    // on the work stack and one partial sum per level on the value stack (16 KiB per block: 9 blocks per CU)
    int w[16][kFeatBlock];     // work stack: pending sub-array lengths, -1 = combine marker
    double v[8][kFeatBlock];   // value stack
};

struct Pairwise {
    double R[8];
    double res, total;
    int rot, cur_len, k, n_left, wsp, vsp;
    PairwiseStacks* st;
    int tid;

    __device__ __forceinline__ void advance() {
        for (;;) {
            if (wsp == 0) {
                if (vsp > 0) { total += st->v[0][tid]; vsp = 0; }  // block done: total += pairwise(block)
                if (n_left == 0) { cur_len = 0; return; }
                const int blk = n_left < 8192 ? n_left : 8192;       // numpy's reduce buffer
                n_left -= blk;
                st->w[wsp++][tid] = blk;
            }
            const int x = st->w[--wsp][tid];
            if (x < 0) {
                const double bb = st->v[--vsp][tid];
                const double aa = st->v[--vsp][tid];
                st->v[vsp++][tid] = aa + bb;
                continue;
            }
            if (x > 128) {
                int n2 = x / 2;
                n2 -= n2 % 8;
                st->w[wsp++][tid] = -1;
                st->w[wsp++][tid] = x - n2;
                st->w[wsp++][tid] = n2;
                continue;
            }
            cur_len = x;
            k = 0;
            return;
        }
    }
    __device__ __forceinline__ void init(int n, int rot_, PairwiseStacks* stacks, int tid_) {
        st = stacks; tid = tid_; rot = rot_ & 7;
        total = 0.0; res = 0.0; n_left = n > 0 ? n : 0; wsp = 0; vsp = 0; cur_len = 0; k = 0;
        advance();
    }
    // r[j] = R[(j + rot) & 7]; ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
    __device__ __forceinline__ double tree() const {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = R[j];
        if (rot & 1) { const double x0 = t[0];
#pragma unroll
            for (int j = 0; j < 7; ++j) t[j] = t[j + 1];
            t[7] = x0; }
        if (rot & 2) { const double x0 = t[0], x1 = t[1];
#pragma unroll
            for (int j = 0; j < 6; ++j) t[j] = t[j + 2];
            t[6] = x0; t[7] = x1; }
        if (rot & 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double x = t[j]; t[j] = t[j + 4]; t[j + 4] = x; }
        }
        return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    // next element of the array; JJ = (its index + rot) & 7, a compile-time chunk position
    template <int JJ>
    __device__ __forceinline__ void feed(double a) {
        if (cur_len < 8) {
            res = (k == 0 ? 0.0 : res) + a;
        } else {
            const int m = cur_len & ~7;
            if (k < 8) R[JJ] = a;
            else if (k < m) R[JJ] += a;
            else { if (k == m) res = tree(); res += a; }
        }
        ++k;
        if (k == cur_len) {
            if (cur_len >= 8 && (cur_len & 7) == 0) res = tree();
            st->v[vsp++][tid] = res;
            advance();
        }
    }
    __device__ __forceinline__ double result() const { return total; }  // 0.0 + block sums, in order
};

__device__ __forceinline__ void py_slice(int64_t start, int64_t end, int has_end, int L, int& lo,
                                         int& hi) {
    int64_t s = start;
    if (s < 0) { s += L; if (s < 0) s = 0; } else if (s > L) s = L;
    int64_t e = has_end ? end : (int64_t)L;
    if (e < 0) { e += L; if (e < 0) e = 0; } else if (e > L) e = L;
    lo = (int)s;
    hi = (int)(e < s ? s : e);
}

// 8 wave values of the aligned chunk c (pool samples 8c..8c+7) as float64 and float32
template <int SRC>
__device__ __forceinline__ void load_chunk(const PoolView& pool, int64_t c, double (&wd)[8], float (&wf)[8]) {
    if (SRC == WFA_SRC_RAW) {
        const uint4 v = reinterpret_cast<const uint4*>(pool.u16)[c];
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t x = (d[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            wf[j] = (float)x;
            wd[j] = (double)x;
        }
    } else {
        const float4 v0 = reinterpret_cast<const float4*>(pool.f32)[2 * c];
        const float4 v1 = reinterpret_cast<const float4*>(pool.f32)[2 * c + 1];
        const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { wf[j] = f[j]; wd[j] = (double)f[j]; }
    }
}

template <int SRC>
__global__ __launch_bounds__(kFeatBlock) void k_basic_features(PoolView pool, RecView rec, FeatParams fp,
                                                               uint8_t* __restrict__ out) {
    __shared__ PairwiseStacks stacks;
    const int64_t r = (int64_t)blockIdx.x * kFeatBlock + threadIdx.x;
    if (r >= rec.R) return;
    const int L = rec.len[r];
    const int64_t off = rec.off[r];
    double baseline = rec.baseline[r];
    if (fp.fixed_bl) {
        const double fb = fp.fixed_bl[r];
        if (fb == fb) baseline = fb;  // basic_features.py:143-146
    }
    const int pol = rec.pol[r];
    const bool known = pol == WFA_POL_NEGATIVE || pol == WFA_POL_POSITIVE;
    const bool wpos = pol == WFA_POL_POSITIVE_WAVE;  // st_waveforms branch, polarity "positive" (basic_features.py:245-262)
    const float b32 = (float)baseline;
    int p0, p1, c0, c1;
    py_slice(fp.h0, fp.h1, fp.h_has_end, L, p0, p1);
    py_slice(fp.a0, fp.a1, fp.a_has_end, L, c0, c1);

    Pairwise pw;
    pw.init(c1 - c0, (int)((off + c0) & 7), &stacks, threadIdx.x);
    const double inf = __builtin_huge_val();
    double vmin = inf, vmax = -inf, mad = 0.0, prev = 0.0;
    if (L > 0) {
        const int64_t c_lo = off >> 3, c_hi = (off + L - 1) >> 3;
        for (int64_t c = c_lo; c <= c_hi; ++c) {
            double wd[8];
            float wf[8];
            load_chunk<SRC>(pool, c, wd, wf);
            const int ib = (int)(c * 8 - off);
#define WFA_BF_STEP(JJ)                                                                              \
            {                                                                                        \
                const int i = ib + JJ;                                                               \
                if (i >= 0 && i < L) {                                                               \
                    double val, term;                                                                \
                    if (known) { /* s = -rv.signals(id, baseline): float32 (records_view.py:87-100) */ \
                        const float dd = wf[JJ] - b32;                                               \
                        val = (double)(pol == WFA_POL_POSITIVE ? dd : -dd);                          \
                        term = val;                                                                  \
                    } else {                                                                         \
                        val = wd[JJ];                                                                \
                        term = wpos ? wd[JJ] - baseline : baseline - wd[JJ]; /* "negative" unless dense-positive */ \
                    }                                                                                \
                    if (i >= p0 && i < p1) { vmin = val < vmin ? val : vmin; vmax = val > vmax ? val : vmax; } \
                    if (i >= c0 && i < c1) pw.feed<JJ>(term);                                        \
                    if (i > 0) { const double dv = fabs(wd[JJ] - prev); mad = dv > mad ? dv : mad; } \
                    prev = wd[JJ];                                                                   \
                }                                                                                    \
            }
            WFA_BF_STEP(0) WFA_BF_STEP(1) WFA_BF_STEP(2) WFA_BF_STEP(3)
            WFA_BF_STEP(4) WFA_BF_STEP(5) WFA_BF_STEP(6) WFA_BF_STEP(7)
#undef WFA_BF_STEP
        }
    }
    float height = 0.f, amp = 0.f, area_f = 0.f, mad_f = 0.f;
    if (p1 > p0) {
        height = known ? (float)vmax : (wpos ? (float)(vmax - baseline) : (float)(baseline - vmin));
        amp = (float)(vmax - vmin);
    }
    if (c1 > c0) area_f = (float)pw.result();
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

template <int SRC>
__global__ __launch_bounds__(kFeatBlock) void k_width_integral(PoolView pool, RecView rec, WidthParams wp,
                                                               uint8_t* __restrict__ out) {
    __shared__ PairwiseStacks stacks;
    // occupancy cap: this kernel walks every record twice and lives on its lanes' cache lines staying in L2 between
    // the 16-byte loads; 6 blocks per CU (24 KiB of LDS each) measured 2.30 ms, 9 blocks 2.43 ms
    __shared__ int occupancy_pad[2048];
    if (threadIdx.x == 0) reinterpret_cast<volatile int*>(occupancy_pad)[0] = 0;
    const int64_t r = (int64_t)blockIdx.x * kFeatBlock + threadIdx.x;
    if (r >= rec.R) return;
    const int L = rec.len[r];
    const int64_t off = rec.off[r];
    const double baseline = rec.baseline[r];
    const float b32 = (float)baseline;
    const int pol = rec.pol[r];
    const bool known = pol == WFA_POL_NEGATIVE || pol == WFA_POL_POSITIVE;
    const bool wpos = pol == WFA_POL_POSITIVE_WAVE;  // dense branch, polarity "positive" (waveform_width_integral.py:184-189)
    const int64_t c_lo = off >> 3, c_hi = (off + L - 1) >> 3;

    // x_i = max(signal_i, 0)   (waveform_width_integral.py:180-190)
    auto xval = [&](double wdj, float wfj) {
        double sgl;
        if (known) {
            const float dd = wfj - b32;
            sgl = (double)(pol == WFA_POL_POSITIVE ? dd : -dd);
        } else {
            sgl = wpos ? wdj - baseline : -(wdj - baseline);
        }
        return sgl > 0.0 ? sgl : 0.0;
    };

    Pairwise pw;
    pw.init(L, (int)(off & 7), &stacks, threadIdx.x);
    if (L > 0) {
        // two chunk buffers: the next chunk is in flight while the current one is summed
        double wda[8], wdb[8];
        float wfa[8], wfb[8];
#define WFA_WI_SUM(WD, WF, JJ) { const int i = ib + JJ; if (i >= 0 && i < L) pw.feed<JJ>(xval(WD[JJ], WF[JJ])); }
#define WFA_WI_CHUNK(WD, WF)                                                                                   \
        {                                                                                                      \
            const int ib = (int)(c * 8 - off);                                                                 \
            WFA_WI_SUM(WD, WF, 0) WFA_WI_SUM(WD, WF, 1) WFA_WI_SUM(WD, WF, 2) WFA_WI_SUM(WD, WF, 3)            \
            WFA_WI_SUM(WD, WF, 4) WFA_WI_SUM(WD, WF, 5) WFA_WI_SUM(WD, WF, 6) WFA_WI_SUM(WD, WF, 7)            \
        }
        load_chunk<SRC>(pool, c_lo, wda, wfa);
        for (int64_t c = c_lo;;) {
            load_chunk<SRC>(pool, c + 1, wdb, wfb);
            WFA_WI_CHUNK(wda, wfa)
            if (++c > c_hi) break;
            load_chunk<SRC>(pool, c + 1, wda, wfa);
            WFA_WI_CHUNK(wdb, wfb)
            if (++c > c_hi) break;
        }
#undef WFA_WI_CHUNK
#undef WFA_WI_SUM
    }
    const double q = pw.result();

    int lo_i = 0, hi_i = 0;
    const bool ok = q > 0.0 && q <= 1.7976931348623157e308;  // finite and positive
    if (ok) {
        const double t_lo = wp.q_low * q, t_hi = wp.q_high * q;
        lo_i = -1; hi_i = -1;
        double cs = 0.0;  // np.cumsum: sequential
        for (int64_t c = c_lo; c <= c_hi && hi_i < 0; ++c) {
            double wd[8];
            float wf[8];
            load_chunk<SRC>(pool, c, wd, wf);
            const int ib = (int)(c * 8 - off);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int i = ib + jj;
                if (i >= 0 && i < L) {
                    cs += xval(wd[jj], wf[jj]);
                    if (lo_i < 0 && cs >= t_lo) lo_i = i;
                    if (hi_i < 0 && cs >= t_hi) hi_i = i;
                }
            }
        }
        if (lo_i < 0) lo_i = L;  // np.searchsorted returns len(cumsum)
        if (hi_i < 0) hi_i = L;
    }
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

// =============================================================================================
// K14: legacy threshold crossings on dense rows  (event_grouping.py:46-95 `find_hits`)
// =============================================================================================
// mask = (baseline - wave) > threshold in float64; one row per wave; a hit = a 0 -> 1 transition of the mask.
// Output (event_index, start sample) in row-major order, which is np.where's order.
template <int SRC, bool FILL>
__global__ __launch_bounds__(kBlock) void k_find_hits_legacy(PoolView pool, int64_t n_rows, int32_t L,
                                                            const double* __restrict__ baselines, double threshold,
                                                            int32_t* __restrict__ counts,
                                                            const int64_t* __restrict__ out_start,
                                                            int64_t* __restrict__ out_event, int64_t* __restrict__ out_time) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (row >= n_rows) return;
    const double b = baselines[row];
    const uint16_t* xu = pool.u16 ? pool.u16 + row * L : nullptr;
    const float* xf = pool.f32 ? pool.f32 + row * L : nullptr;
    int64_t n_out = FILL ? out_start[row] : 0;
    uint64_t prev_last = 0;  // mask bit of the sample before this 64-sample block
    for (int base = 0; base < L; base += kWave) {
        const int i = base + lane;
        bool m = false;
        if (i < L) {
            const double w = SRC == WFA_SRC_RAW ? (double)xu[i] : (double)xf[i];
            m = (b - w) > threshold;
        }
        const uint64_t bits = __ballot(m);
        const uint64_t starts = bits & ~((bits << 1) | prev_last);
        if (FILL && ((starts >> lane) & 1ull)) {
            const int64_t k = n_out + __popcll(starts & ((1ull << lane) - 1ull));
            out_event[k] = row;
            out_time[k] = i;
        }
        n_out += __popcll(starts);
        prev_last = bits >> 63;
    }
    if (!FILL && lane == 0) counts[row] = (int32_t)n_out;
}

// =============================================================================================
// K10: rise / fall / total width per hit on dense rows  (waveform_width.py:205-374)
// =============================================================================================
// One lane per hit.  T = double for int16 rows (numpy promotes int16 - float64 mean to float64), float for
// float32 rows (mean, subtraction, thresholds, interpolation all stay float32 under numpy 2 promotion).
// numpy float32 pairwise_sum of x[0..n), n <= 128 (np.mean / np.sum of a short float32 slice)
__device__ inline float np_pairwise_leaf_f32(const float* x, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res += x[i];
        return res;
    }
    float r0 = x[0], r1 = x[1], r2 = x[2], r3 = x[3], r4 = x[4], r5 = x[5], r6 = x[6], r7 = x[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += x[i]; r1 += x[i + 1]; r2 += x[i + 2]; r3 += x[i + 3];
        r4 += x[i + 4]; r5 += x[i + 5]; r6 += x[i + 6]; r7 += x[i + 7];
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += x[i];
    return res;
}

struct WidthHitParams {
    double rise_low, rise_high, fall_high, fall_low, sampling_rate;
    int interpolation;
    int32_t L;
    int64_t n_rows;
};

// value that is either a python float / np.float64 (is32 = false) or an np.float32 (is32 = true)
struct PyNum {
    double v;
    bool is32;
    bool none;
};

template <int SRC, typename T, typename LoadF>
__device__ PyNum find_crossing(const PoolView& pool, int64_t base, T baseline, const LoadF& at, int lo, int hi, T thr,
                               bool rising, bool interp) {
    PyNum out{0.0, false, true};
    // first index (relative to lo) whose value is >= thr (rising) / <= thr (falling).  The search reads aligned
    // 16-byte chunks and tests their samples from registers: with one load per step every step waited for memory.
    int idx = -1;
    {
        const int64_t g_lo = base + lo, g_hi = base + hi;  // pool positions [g_lo, g_hi)
        bool found = false;
        for (int64_t c = g_lo >> 3; !found && c * 8 < g_hi; ++c) {
            double wd[8];
            float wf[8];
            load_chunk<SRC>(pool, c, wd, wf);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t g = c * 8 + j;
                if (!found && g >= g_lo && g < g_hi) {
                    const T y = SRC == WFA_SRC_RAW ? (T)(wd[j] - (double)baseline) : (T)(wf[j] - (float)baseline);
                    if (rising ? (y >= thr) : (y <= thr)) { idx = (int)(g - g_lo); found = true; }
                }
            }
        }
    }
    if (idx < 0) return out;
    out.none = false;
    out.v = (double)idx;  // float(idx): python float
    if (!interp || idx == 0) return out;
    const T y0 = at(lo + idx - 1), y1 = at(lo + idx);
    const T d = y1 - y0;
    const T ad = d < (T)0 ? -d : d;
    if (ad < (T)1e-10) return out;  // python float 1e-10 is weak: compared in T
    const T fraction = (thr - y0) / d;
    const T pos = (T)(idx - 1) + fraction;  // float(idx - 1) + fraction in T
    out.v = (double)pos;
    out.is32 = sizeof(T) == 4;
    return out;
}

__device__ __forceinline__ PyNum py_sub(const PyNum& a, const PyNum& b) {
    PyNum r{0.0, a.is32 || b.is32, false};
    r.v = r.is32 ? (double)((float)a.v - (float)b.v) : a.v - b.v;
    return r;
}

__device__ __forceinline__ PyNum py_div(const PyNum& a, double python_float) {
    PyNum r{0.0, a.is32, false};
    r.v = a.is32 ? (double)((float)a.v / (float)python_float) : a.v / python_float;
    return r;
}

template <int SRC>
__global__ __launch_bounds__(128) void k_waveform_width(PoolView pool, int64_t n_hits,
                                                        const int64_t* __restrict__ position,
                                                        const int64_t* __restrict__ row_index, WidthHitParams wp,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ valid) {
    using T = typename std::conditional<SRC == WFA_SRC_RAW, double, float>::type;
    const int64_t h = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (h >= n_hits) return;
    uint32_t* row = reinterpret_cast<uint32_t*>(out + h * 56);
#pragma unroll
    for (int k = 0; k < 14; ++k) row[k] = 0u;
    valid[h] = 0;
    const int64_t ri = row_index[h];
    const int64_t pos64 = position[h];
    const int L = wp.L;
    if (ri < 0 || ri >= wp.n_rows || L <= 0) return;
    if (pos64 >= L || pos64 < 0) return;  // waveform_width.py:252
    const int peak = (int)pos64;
    const uint16_t* xu = pool.u16 ? pool.u16 + ri * L : nullptr;
    const float* xf = pool.f32 ? pool.f32 + ri * L : nullptr;

    // baseline = np.mean(waveform[:50])
    const int nb = L < 50 ? L : 50;
    T baseline;
    if (SRC == WFA_SRC_RAW) {
        double sum = 0.0;  // integers: exact in float64 whatever the order
        for (int i = 0; i < nb; ++i) sum += (double)xu[i];
        baseline = (T)(sum / (double)nb);
    } else {
        const float sum = np_pairwise_leaf_f32(xf, nb);
        baseline = (T)(sum / (float)nb);
    }
    auto at = [&](int i) -> T {
        if (SRC == WFA_SRC_RAW) return (T)((double)xu[i] - (double)baseline);
        return (T)(xf[i] - (float)baseline);
    };
    const T peak_value = at(peak);
    if (!(peak_value > (T)0)) return;  // waveform_width.py:258 (NaN also compares false there -> kept; NaN is not produced)
    const bool interp = wp.interpolation != 0;
    // python float options are weak: thresholds take the dtype of peak_value
    const T thr_rl = peak_value * (T)wp.rise_low, thr_rh = peak_value * (T)wp.rise_high;
    const T thr_fh = peak_value * (T)wp.fall_high, thr_fl = peak_value * (T)wp.fall_low;
    const int64_t base = ri * L;
    const PyNum rise_lo = find_crossing<SRC, T>(pool, base, baseline, at, 0, peak, thr_rl, true, interp);
    const PyNum rise_hi = find_crossing<SRC, T>(pool, base, baseline, at, 0, peak, thr_rh, true, interp);
    PyNum fall_hi = find_crossing<SRC, T>(pool, base, baseline, at, peak, L, thr_fh, false, interp);
    PyNum fall_lo = find_crossing<SRC, T>(pool, base, baseline, at, peak, L, thr_fl, false, interp);

    PyNum rise_s{0.0, false, false}, rise_t{0.0, false, false};
    if (!rise_lo.none && !rise_hi.none) {
        rise_s = py_sub(rise_hi, rise_lo);
        rise_t = py_div(rise_s, wp.sampling_rate);
    }
    PyNum fall_s{0.0, false, false}, fall_t{0.0, false, false};
    if (!fall_hi.none && !fall_lo.none) {
        // += np.int64 peak_position promotes to float64 whatever the left side was
        fall_hi.v = fall_hi.v + (double)peak; fall_hi.is32 = false;
        fall_lo.v = fall_lo.v + (double)peak; fall_lo.is32 = false;
        fall_s = py_sub(fall_lo, fall_hi);
        fall_t = py_div(fall_s, wp.sampling_rate);
    }
    PyNum tot_s{0.0, false, false}, tot_t{0.0, false, false};
    if (!rise_lo.none && !fall_lo.none) {
        // fall_low_pos is float64 only if the fall branch above ran (it needs fall_high_pos too)
        if (!fall_lo.is32 && (!fall_hi.none)) {
            tot_s.v = fall_lo.v - rise_lo.v;  // np.float64 - (np.float32 | python float) -> float64
            tot_s.is32 = false;
        } else {
            tot_s = py_sub(fall_lo, rise_lo);
        }
        tot_t = py_div(tot_s, wp.sampling_rate);
    }
    put_f32(row, 0, (float)rise_t.v);
    put_f32(row, 1, (float)fall_t.v);
    put_f32(row, 2, (float)tot_t.v);
    put_f32(row, 3, (float)rise_s.v);
    put_f32(row, 4, (float)fall_s.v);
    put_f32(row, 5, (float)tot_s.v);
    put_i64(row, 6, pos64);
    put_f32(row, 8, (float)peak_value);
    valid[h] = 1;
}

// =============================================================================================
// K8: find_peaks-based hit detector  (peak_finding.py:395-614, records branch; scipy.signal.find_peaks)
// =============================================================================================
// One lane per record executes scipy's algorithm literally on the detection signal
//   det = diff(signal)  (use_derivative)  or  signal,   signal = -rv.signals(id) as float64 of float32
// in scipy's order: local maxima with plateaus (_local_maxima_1d, streamed) -> height -> threshold ->
// distance (_select_by_peak_distance) -> prominence (_peak_prominences) -> width at half prominence
// (_peak_widths, interpolated intersection points).  Passes: candidates per record (count, scan, fill),
// k_peak_select (distance > 2 only; a no-op for local maxima otherwise), then one lane per candidate for the
// prominence / width walks and, after a scan of the accept flags, for the rows: (record, position) order without a
// gather, nothing capped.
constexpr int kPeakBlock = 128;
constexpr int kPeakErrEmptyWindow = 2;

// MODE >= 0 fixes the form of the detection value at compile time (the kernels that walk a lot dispatch on it once per
// launch: the tests inside det_of were most of their scalar instruction stream); -1 = decided per call.
//   0 records, derivative   1 records, plain   2 rows, plain   3 rows, derivative in float64   4 rows, derivative in float32
template <int SRC, int MODE = -1>
struct SignalAt {
    const uint16_t* xu;
    const float* xf;
    float b32;
    double b64;
    bool positive;
    int use_derivative;
    int rows;  // WFA_PEAK_SIGNAL_ROWS / _ROWS_F64: the stored samples are the waveform (negative-going pulses)
    int L, n;  // samples in the record, samples of the detection signal
    int64_t off0;  // the record's first sample in the pool
    PoolView pv;
    // the waveform the height is measured on
    __device__ __forceinline__ double sig(int i) const {
        const float w = SRC == WFA_SRC_RAW ? (float)xu[i] : xf[i];
        if (rows) return (double)w;
        const float d = w - b32;             // records_view.py:87-100 (float32)
        return (double)(positive ? d : -d);  // signal = -normalized
    }
    // value k of the detection signal from the float32 samples w0 = w[k], w1 = w[k + 1]  (peak_finding.py:490-510)
    __device__ __forceinline__ int deriv() const {
        if constexpr (MODE >= 0) return MODE == 0 || MODE == 3 || MODE == 4;
        else return use_derivative;
    }
    __device__ __forceinline__ double det_of(float w0, float w1) const {
        if constexpr (MODE == 0 || MODE == 1) {
            const float d0 = w0 - b32;
            const double s0 = (double)(positive ? d0 : -d0);
            if constexpr (MODE == 1) return s0 - 0.0;
            const float d1 = w1 - b32;
            return (double)(positive ? d1 : -d1) - s0;
        } else if constexpr (MODE == 2) {
            return b64 - (double)w0;
        } else if constexpr (MODE == 3) {
            return -((double)w1 - (double)w0);
        } else if constexpr (MODE == 4) {
            return (double)(-(w1 - w0));
        }
        if (rows) {
            if (!use_derivative) return b64 - (double)w0;                   // np.float64 baseline - row
            // -np.diff(int16 row) is exact; the streaming detector converts the row to float64 first
            if (SRC == WFA_SRC_RAW || rows == WFA_PEAK_SIGNAL_ROWS_F64) return -((double)w1 - (double)w0);
            return (double)(-(w1 - w0));                                    // -np.diff(float32 row): float32
        }
        const float d0 = w0 - b32;
        const double s0 = (double)(positive ? d0 : -d0);
        if (!use_derivative) return s0 - 0.0;
        const float d1 = w1 - b32;
        return (double)(positive ? d1 : -d1) - s0;
    }
    __device__ __forceinline__ float wave(int i) const { return SRC == WFA_SRC_RAW ? (float)xu[i] : xf[i]; }
    __device__ __forceinline__ double det(int i) const {
        return det_of(wave(i), deriv() ? wave(i + 1) : 0.f);
    }
    __device__ __forceinline__ void bind(const PoolView& pool, const RecView& rec, int64_t r, const PeakParams& pp) {
        const int64_t off = rec.off[r];
        off0 = off;
        pv = pool;
        xu = pool.u16 ? pool.u16 + off : nullptr;
        xf = pool.f32 ? pool.f32 + off : nullptr;
        b64 = rec.baseline[r];
        b32 = (float)b64;
        positive = rec.pol[r] == WFA_POL_POSITIVE;
        use_derivative = pp.use_derivative;
        rows = pp.rows;
        L = rec.len[r];
        n = use_derivative ? L - 1 : L;
    }
};

template <int SRC>
__device__ __host__ inline int signal_mode(int rows, int use_derivative) {
    if (!rows) return use_derivative ? 0 : 1;
    if (!use_derivative) return 2;
    return (SRC == WFA_SRC_RAW || rows == WFA_PEAK_SIGNAL_ROWS_F64) ? 3 : 4;
}

// det(i) for i = i_start, i_start + DIR, ... while the visitor returns true and i stays in [0, n).  The samples come
// in aligned 16-byte chunks (8 uint16, or 2 x 4 float32) and are consumed from registers with static indices: the walks
// below are pointer chases of up to a whole record per candidate, and with one load per step each step waited for a
// cache line (5.1 ms per 10^9 samples for k_peak_eval; chunked: see DESIGN.md).  A derivative value needs the sample
// after it, which is carried from the previous step (DIR < 0) or makes the value one step late (DIR > 0).
template <int SRC, int DIR, typename SigT, typename F>
__device__ __forceinline__ void stream_det(const SigT& S, int i_start, const F& visit) {
    const int deriv = S.deriv();
    // sample index range this walk reads, in walking order
    const int k0 = DIR > 0 ? i_start : i_start + deriv;
    if (k0 < 0 || k0 >= S.L) return;
    float carry = 0.f;
    bool have_carry = false, active = true;
    int64_t c = (S.off0 + k0) >> 3;
    const int64_t c_lo = S.off0 >> 3, c_hi = (S.off0 + S.L - 1) >> 3;
    auto consume_some = [&](int64_t cq, const float (&wf)[8], auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;  // every sample of the chunk is part of the walk
        const int kb = (int)(cq * 8 - S.off0);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = DIR > 0 ? jj : 7 - jj;
            const int k = kb + j;
            const bool in = FULL || (DIR > 0 ? (k >= k0 && k < S.L) : (k <= k0 && k >= 0));  // false for a chunk outside the record
            if (active && in) {
                const float w = wf[j];
                if (!deriv) {
                    active = visit(k, S.det_of(w, 0.f));
                } else if (DIR > 0) {
                    if (FULL || have_carry) active = visit(k - 1, S.det_of(carry, w));
                    carry = w; have_carry = true;
                } else {
                    if (FULL || have_carry) active = visit(k, S.det_of(w, carry));
                    carry = w; have_carry = true;
                }
            }
        }
    };
    // Records that start and end on chunk boundaries (every uniform layout) have a partial chunk only where the walk
    // begins: from the second step on no sample needs its range test (`full` is then true in every lane of the wave)
    auto consume = [&](int64_t cq, const float (&wf)[8]) __attribute__((always_inline)) {
        const int kb = (int)(cq * 8 - S.off0);
        const bool full = (!deriv || have_carry) && (DIR > 0 ? (kb >= k0 && kb + 7 < S.L) : (kb + 7 <= k0 && kb >= 0));
        if (__ballot(active && !full) == 0) consume_some(cq, wf, std::true_type{});
        else consume_some(cq, wf, std::false_type{});
    };
    // Two chunks (16 samples) per step, the next two requested before these are consumed: every step of this walk used to
    // wait for its own cache line, and a launch of k_peak_eval is as long as its longest chain of such waits (all of its
    // waves are resident at once).  Chunks past the record's ends are read from the clamped address and masked.
    auto pick = [&](int64_t q) { return q < c_lo ? c_lo : (q > c_hi ? c_hi : q); };
    float a0[8], a1[8], b0[8], b1[8];
    double wd[8];
    load_chunk<SRC>(S.pv, pick(c), wd, a0);
    load_chunk<SRC>(S.pv, pick(c + DIR), wd, a1);
    for (;;) {
        load_chunk<SRC>(S.pv, pick(c + 2 * DIR), wd, b0);
        load_chunk<SRC>(S.pv, pick(c + 3 * DIR), wd, b1);
        consume(c, a0);
        consume(c + DIR, a1);
        c += 2 * DIR;
        if (!active || c < c_lo || c > c_hi) break;
        load_chunk<SRC>(S.pv, pick(c + 2 * DIR), wd, a0);
        load_chunk<SRC>(S.pv, pick(c + 3 * DIR), wd, a1);
        consume(c, b0);
        consume(c + DIR, b1);
        c += 2 * DIR;
        if (!active || c < c_lo || c > c_hi) break;
    }
}

// prominence + width of one candidate; false when it fails `prominence` or `width`
// (scipy _peak_prominences / _peak_widths with wlen = None, rel_height = 0.5)
template <int SRC, typename SigT>
__device__ bool peak_passes(const SigT& S, int peak, const PeakParams& pp, double& left_ip, double& right_ip) {
    const double xp = S.det(peak);
    int left_base = peak, right_base = peak;
    double left_min = xp, right_min = xp;
    stream_det<SRC, -1>(S, peak, [&](int i, double v) {
        if (!(v <= xp)) return false;
        if (v < left_min) { left_min = v; left_base = i; }
        return true;
    });
    stream_det<SRC, +1>(S, peak, [&](int i, double v) {
        if (!(v <= xp)) return false;
        if (v < right_min) { right_min = v; right_base = i; }
        return true;
    });
    const double prom = xp - (left_min > right_min ? left_min : right_min);
    if (!(prom >= pp.pmin)) return false;
    const double hgt = xp - prom * 0.5;  // rel_height = 0.5
    {
        // i = peak; while (left_base < i && hgt < x[i]) --i;  then interpolate between x[i] and x[i + 1]
        int i_fin = peak;
        double xi = xp, above = xp;
        stream_det<SRC, -1>(S, peak, [&](int i, double v) {
            i_fin = i; xi = v;
            if (left_base < i && hgt < v) { above = v; return true; }
            return false;
        });
        left_ip = (double)i_fin;
        if (xi < hgt) left_ip += (hgt - xi) / (above - xi);
    }
    {
        int i_fin = peak;
        double xi = xp, above = xp;
        stream_det<SRC, +1>(S, peak, [&](int i, double v) {
            i_fin = i; xi = v;
            if (i < right_base && hgt < v) { above = v; return true; }
            return false;
        });
        right_ip = (double)i_fin;
        if (xi < hgt) right_ip -= (hgt - xi) / (above - xi);
    }
    return right_ip - left_ip >= pp.wmin;
}

// HIT_DTYPE row of one accepted peak (peak_finding.py:508-563 and _calculate_peak_height 567-614)
template <int SRC>
__device__ void write_peak_row(const SignalAt<SRC>& S, const RecView& rec, int64_t r, int peak, double left_ip,
                               double right_ip, const PeakParams& pp, uint32_t* row, int* err, double* pw_scratch) {
    const int L = S.L;
    int start_idx = (int)rint(left_ip), end_idx = (int)rint(right_ip);  // np.round: half to even
    if (start_idx < 0) start_idx = 0;
    if (end_idx > L - 1) end_idx = L - 1;
    double ph;
    if (pp.height_diff) {
        ph = 0.0;
        if (end_idx > start_idx) {
            if (S.rows == WFA_PEAK_SIGNAL_ROWS_F64) {
                // signal_peaks.py:365-375: cumsum(-diff(row))[end] - cumsum[start], np.cumsum is sequential
                double acc = 0.0, at_start = 0.0;
                for (int q = 0; q < end_idx; ++q) {
                    if (q == start_idx) at_start = acc;
                    acc += -(S.sig(q + 1) - S.sig(q));
                }
                ph = (double)(float)(acc - at_start);  // heights.astype(np.float32)
            } else if (S.rows && SRC == WFA_SRC_F32)  // np.sum(np.diff(-row)) of a float32 row stays float32
                ph = (double)np_pairwise_sum<float>([&](int q) { return (-S.xf[q + 1]) - (-S.xf[q]); }, start_idx,
                                                    end_idx - start_idx, pw_scratch, kPeakBlock);
            else
                ph = np_pairwise_sum([&](int q) { return (-S.sig(q + 1)) - (-S.sig(q)); }, start_idx,
                                     end_idx - start_idx, pw_scratch, kPeakBlock);
        }
    } else {
        int w0 = start_idx - pp.ext, w1 = end_idx + pp.ext;
        if (w0 < 0) w0 = 0;
        if (w1 > L) w1 = L;
        if (w1 <= w0) atomicExch(err, kPeakErrEmptyWindow);  // numpy: max of a zero-size array
        double vmax = -__builtin_huge_val(), vmin = __builtin_huge_val();
        for (int q = w0; q < w1; ++q) {
            const double v = S.sig(q);
            vmax = v > vmax ? v : vmax;
            vmin = v < vmin ? v : vmin;
        }
        ph = S.rows == WFA_PEAK_SIGNAL_ROWS && SRC == WFA_SRC_F32 ? (double)((float)vmax - (float)vmin) : vmax - vmin;
    }
    const int dt_ns = rec.dt[r];
    put_i64(row, 0, (int64_t)peak);
    put_f32(row, 2, (float)ph);
    put_f32(row, 3, 0.0f);
    put_f32(row, 4, (float)left_ip);
    put_f32(row, 5, (float)right_ip);
    row[6] = (uint32_t)dt_ns;
    put_i64(row, 7, (int64_t)((double)rec.ts[r] + (double)peak * ((double)dt_ns * 1e3)));
    row[9] = (uint32_t)(uint16_t)rec.board[r] | ((uint32_t)(uint16_t)rec.chan[r] << 16);
    put_i64(row, 10, rec.rid[r]);
}

// Streaming _local_maxima_1d + height + threshold; calls on_peak(position, value) per candidate in order.
template <int SRC, typename SigT, typename F>
__device__ void scan_candidates(const PoolView& pool, int64_t off, const SigT& S, const PeakParams& pp,
                                const F& on_peak) {
    const int n = S.n;
    if (n < 3) return;
    const int L = S.L;
    bool have = false;
    int c_start = 0;
    double c_val = 0.0, x_prev = 0.0;
    float w_prev = 0.f;
    // one detection value per step; i is its index.  Samples arrive in aligned 16-byte chunks (8 uint16 or
    // 2 x 4 float32 per lane and load): a lane walks its own record, so single-sample loads would pull a whole
    // cache line per 2..4 useful bytes (measured 26 ms per 10^9 samples for the count pass alone).
    auto step = [&](int i, double x) {
        if (i == 0) { x_prev = x; return; }
        if (have) {
            if (x < c_val) {
                const int peak = (c_start + i - 1) / 2;  // midpoint of the plateau
                bool keep = c_val >= pp.hmin;
                if (keep && pp.has_threshold) {
                    const double lt = c_val - S.det(peak - 1), rt = c_val - S.det(peak + 1);
                    keep = (lt < rt ? lt : rt) >= pp.tmin;
                }
                if (keep) on_peak(peak, c_val);
                have = false;
            } else if (x > c_val) {
                c_start = i; c_val = x;  // a higher step starts a new candidate
            }
        } else if (x_prev < x && i < n - 1) {
            have = true; c_start = i; c_val = x;
        }
        x_prev = x;
    };
    const int64_t c_lo = off >> 3, c_hi = (off + L - 1) >> 3;
    auto consume = [&](int64_t c, const float (&wf)[8]) {
        const int kb = (int)(c * 8 - off);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = kb + jj;  // sample index in the record
            if (k < 0 || k >= L) continue;
            if (S.deriv()) {
                if (k >= 1) step(k - 1, S.det_of(w_prev, wf[jj]));
                w_prev = wf[jj];
            } else {
                step(k, S.det_of(wf[jj], 0.f));
            }
        }
    };
    // the next chunk is requested before the current one is consumed (two buffers, loop unrolled by two so that no
    // register copy waits for the load); the chunk behind the record's last one is a valid address (pool slack).
    // (A ring of four, unrolled by four, makes the body large enough that hipcc stops inlining it: 3x slower.)
    float wa[8], wb[8];
    double wd[8];
    load_chunk<SRC>(pool, c_lo, wd, wa);
    for (int64_t c = c_lo;;) {
        load_chunk<SRC>(pool, c + 1, wd, wb);
        consume(c, wa);
        if (++c > c_hi) break;
        load_chunk<SRC>(pool, c + 1, wd, wa);
        consume(c, wb);
        if (++c > c_hi) break;
    }
}

// Candidate pass: FILL = false counts the local maxima that pass `height` / `threshold` per record, FILL = true
// writes (position, value, record) at the scanned offsets.  Everything after it runs one lane per CANDIDATE:
// the prominence / width walks are data-dependent pointer walks, and inside the per-record scan they ran with the
// other 63 lanes of the wave idle (36 ms per 10^9 samples against 2 ms for the scan itself).
template <int SRC, bool FILL>
__global__ __launch_bounds__(kPeakBlock) void k_find_peaks(PoolView pool, RecView rec, PeakParams pp,
                                                           int32_t* __restrict__ counts,
                                                           const int64_t* __restrict__ out_start,
                                                           int32_t* __restrict__ cand_pos, double* __restrict__ cand_val,
                                                           int64_t* __restrict__ cand_rec) {
    const int64_t r = (int64_t)blockIdx.x * kPeakBlock + threadIdx.x;
    if (r >= rec.R) return;
    int n_out = 0;
    auto run = [&](auto mode_tag) __attribute__((always_inline)) {
        SignalAt<SRC, decltype(mode_tag)::value> S;
        S.bind(pool, rec, r, pp);
        if (S.L > 0) {
            const int64_t base = FILL ? out_start[r] : 0;
            scan_candidates<SRC>(pool, rec.off[r], S, pp, [&](int peak, double val) {
                if (FILL) {
                    cand_pos[base + n_out] = peak;
                    cand_val[base + n_out] = val;
                    cand_rec[base + n_out] = r;
                }
                ++n_out;
            });
        }
    };
    switch (signal_mode<SRC>(pp.rows, pp.use_derivative)) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        default: run(std::integral_constant<int, 4>{}); break;
    }
    if (!FILL) counts[r] = n_out;
}

// One walk instead of count + fill: every record writes its first `K` candidates into its own K slots and its full
// count; the scan of the counts gives the compact offsets and k_peak_compact moves the slots there.  A record with
// more than K candidates raises `overflow` and the caller falls back to the fill walk (the counts are exact either way).
template <int SRC>
__global__ __launch_bounds__(kPeakBlock) void k_find_peaks_slots(PoolView pool, RecView rec, PeakParams pp, int K,
                                                                 int32_t* __restrict__ counts,
                                                                 int32_t* __restrict__ slot_pos,
                                                                 double* __restrict__ slot_val, int* __restrict__ overflow) {
    const int64_t r = (int64_t)blockIdx.x * kPeakBlock + threadIdx.x;
    if (r >= rec.R) return;
    int n_out = 0;
    auto run = [&](auto mode_tag) __attribute__((always_inline)) {
        SignalAt<SRC, decltype(mode_tag)::value> S;
        S.bind(pool, rec, r, pp);
        if (S.L > 0) {
            const int64_t base = r * K;
            scan_candidates<SRC>(pool, rec.off[r], S, pp, [&](int peak, double val) {
                if (n_out < K) {
                    slot_pos[base + n_out] = peak;
                    slot_val[base + n_out] = val;
                }
                ++n_out;
            });
        }
    };
    switch (signal_mode<SRC>(pp.rows, pp.use_derivative)) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        default: run(std::integral_constant<int, 4>{}); break;
    }
    counts[r] = n_out;
    if (n_out > K) atomicOr(overflow, 1);
}

// The same single walk for uniform records, with coalesced reads: a wave stages a group of 64 >> gl_shift consecutive
// records in LDS (16-byte loads, every byte of the pool read once) and 1 << gl_shift lanes share a record, each scanning
// its own stretch of the detection signal.  _local_maxima_1d is local: a candidate is a rise, a plateau and a fall, so a
// lane that starts with no open candidate at its first index finds exactly the candidates whose plateau STARTS in its
// stretch, provided it follows an open one past the end of the stretch until it falls (a peak) or rises (the later
// plateau start belongs to the lane behind).  Lanes are in position order, so are the record's candidates.
struct StagedPeakArgs {
    int64_t off0;      // first sample of record 0 in the pool
    int32_t L;         // samples per record (multiple of 8)
    int32_t gl_shift;  // lanes per record = 1 << gl_shift
};

template <int SRC>
__global__ __launch_bounds__(kWave) void k_find_peaks_staged(PoolView pool, RecView rec, PeakParams pp, StagedPeakArgs sa,
                                                             int K, int32_t* __restrict__ counts,
                                                             int32_t* __restrict__ slot_pos, double* __restrict__ slot_val,
                                                             int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    constexpr int ES = SRC == WFA_SRC_RAW ? 2 : 4;
    const int lane = lane_id();
    const int L = sa.L;
    const int GL = 1 << sa.gl_shift, RW = kWave >> sa.gl_shift;
    const int g = lane >> sa.gl_shift, k = lane & (GL - 1);
    // all of a group's loads (samples and record columns) are issued before anything waits for one of them; the loop
    // runs once per wave with the launcher's grid (it is written for any grid)
    const int64_t n_groups = (rec.R + RW - 1) / RW;
    const u4* __restrict__ src0 = reinterpret_cast<const u4*>(SRC == WFA_SRC_RAW ? (const void*)pool.u16 : (const void*)pool.f32) +
                                  (sa.off0 * ES >> 4);
    const int gchunks = L * ES >> 4;  // 16-byte chunks per record
    u4 pf[16];
    double nx_bl = 0.0;
    int nx_pol = 0;
    auto fetch = [&](int64_t grp) __attribute__((always_inline)) {
        const int64_t q0 = grp * RW;
        const int nr = (int)(rec.R - q0 < RW ? rec.R - q0 : RW);
        const int chunks = nr * gchunks;  // <= 1024: 16 KiB per wave
        const u4* src = src0 + q0 * gchunks;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = t * kWave + lane;
            pf[t] = src[c < chunks ? c : 0];
        }
        const int64_t q = g < nr ? q0 + g : q0;
        nx_bl = rec.baseline[q];
        nx_pol = rec.pol[q];
    };
    int64_t grp = blockIdx.x;
    if (grp < n_groups) fetch(grp);
    for (; grp < n_groups; grp += gridDim.x) {
    const int64_t r0 = grp * RW;
    const int nrec = (int)(rec.R - r0 < RW ? rec.R - r0 : RW);
    {
        const int chunks = nrec * gchunks;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = t * kWave + lane;
            if (c < chunks) reinterpret_cast<u4*>(s_stage)[c] = pf[t];
        }
    }
    const double me_bl = nx_bl;
    const int me_pol = nx_pol;
    if (grp + gridDim.x < n_groups) fetch(grp + gridDim.x);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const bool valid = g < nrec;
    const int64_t r = valid ? r0 + g : r0;
    SignalAt<SRC> S;
    {
        const int64_t off = sa.off0 + r * L;  // uniform records: nothing to load
        S.off0 = off;
        S.pv = pool;
        S.xu = pool.u16 ? pool.u16 + off : nullptr;
        S.xf = pool.f32 ? pool.f32 + off : nullptr;
        S.b64 = me_bl;
        S.b32 = (float)me_bl;
        S.positive = me_pol == WFA_POL_POSITIVE;
        S.use_derivative = pp.use_derivative;
        S.rows = pp.rows;
        S.L = L;
        S.n = pp.use_derivative ? L - 1 : L;
    }
    const int n = S.n, deriv = S.use_derivative;
    const uint16_t* lu = reinterpret_cast<const uint16_t*>(s_stage) + g * L;
    const float* lf = reinterpret_cast<const float*>(s_stage) + g * L;
    auto wv = [&](int m) { return SRC == WFA_SRC_RAW ? (float)lu[m] : lf[m]; };

    constexpr int kStash = 4;  // candidates a lane keeps (ringing behind a large pulse gives a lane three or four)
    int n_mine = 0;
    int st_pos[kStash] = {0, 0, 0, 0};
    double st_val[kStash] = {0.0, 0.0, 0.0, 0.0};
    auto on_peak = [&](int c_start, int i, double c_val) {  // plateau [c_start, i - 1] ended by a fall at i
        const int peak = (c_start + i - 1) / 2;
        bool keep = c_val >= pp.hmin;
        if (keep && pp.has_threshold) {
            const double lt = c_val - S.det(peak - 1), rt = c_val - S.det(peak + 1);
            keep = (lt < rt ? lt : rt) >= pp.tmin;
        }
        if (keep) {
#pragma unroll
            for (int q = 0; q < kStash; ++q)
                if (n_mine == q) { st_pos[q] = peak; st_val[q] = c_val; }
            ++n_mine;
        }
    };
    // The detection value in one of five wave-uniform forms (SignalAt::det_of): chosen once per group, so that the walk
    // below is compiled per form and carries no mode tests (they were 54 scalar instructions and 15 branches per step).
    auto walk = [&](auto det2, auto deriv_tag) __attribute__((always_inline)) {
        constexpr int DERIV = decltype(deriv_tag)::value;
        if (!(valid && n >= 3)) return;
        int seg = (n + GL - 1) >> sa.gl_shift;
        seg = (seg + 3) & ~3;
        const int a = k * seg, b = a + seg < n ? a + seg : n;
        if (a >= b) return;
        bool have = false;
        int c_start = 0;
        double c_val = 0.0;
        double x_prev = a > 0 ? det2(wv(a - 1), DERIV ? wv(a) : 0.f) : 0.0;
        float w_prev = wv(a);
        // the plateau state machine of scan_candidates as selects: one (rare) divergent block per step, for the lanes
        // whose candidate ends at this step
        for (int i0 = a; i0 < b; i0 += 4) {
            float wn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = i0 + u + DERIV;
                wn[u] = wv(m < L ? m : L - 1);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                const double x = DERIV ? det2(w_prev, wn[u]) : det2(wn[u], 0.f);
                w_prev = DERIV ? wn[u] : w_prev;
                const bool in = i < b;
                const bool act = in && i > 0;
                const bool fall = act && have && x < c_val;
                if (fall) on_peak(c_start, i, c_val);
                const bool set = act && (have ? x > c_val : (x_prev < x && i < n - 1));
                have = (have && !fall) || (act && set);
                c_start = set ? i : c_start;
                c_val = set ? x : c_val;
                x_prev = in ? x : x_prev;
            }
        }
        for (int i = b; __ballot(have && i < n) != 0; ++i) {  // an open candidate is followed to its end
            const bool go = have && i < n;
            const int m = go ? i : 0;
            const double x = det2(wv(m), DERIV ? wv(m + 1 < L ? m + 1 : L - 1) : 0.f);
            if (go && x < c_val) on_peak(c_start, i, c_val);
            have = go && x == c_val;  // a fall ends it as a peak, a rise as the start of the next lane's plateau
        }
    };
    {
        const float b32 = S.b32;
        const double b64 = S.b64;
        const uint32_t sgn = S.positive ? 0u : 0x80000000u;  // signal = +-(w - baseline) in float32
        auto rec_val = [=](float w) { return (double)__uint_as_float(__float_as_uint(w - b32) ^ sgn); };
        if (!S.rows) {
            if (deriv) walk([=](float w0, float w1) { return rec_val(w1) - rec_val(w0); }, std::integral_constant<int, 1>{});
            else walk([=](float w0, float) { return rec_val(w0) - 0.0; }, std::integral_constant<int, 0>{});
        } else if (!deriv) {
            walk([=](float w0, float) { return b64 - (double)w0; }, std::integral_constant<int, 0>{});
        } else if (SRC == WFA_SRC_RAW || S.rows == WFA_PEAK_SIGNAL_ROWS_F64) {
            walk([=](float w0, float w1) { return -((double)w1 - (double)w0); }, std::integral_constant<int, 1>{});
        } else {
            walk([=](float w0, float w1) { return (double)(-(w1 - w0)); }, std::integral_constant<int, 1>{});
        }
    }
    // slots of the record in lane order
    int incl = n_mine;
    for (int d = 1; d < GL; d <<= 1) {
        const int o = __shfl_up(incl, d, GL);
        if (k >= d) incl += o;
    }
    const int base = incl - n_mine;
    const int total = __shfl(incl, GL - 1, GL);
    if (valid) {
        const int64_t sb = r * K;
#pragma unroll
        for (int q = 0; q < kStash; ++q)
            if (n_mine > q && base + q < K) { slot_pos[sb + base + q] = st_pos[q]; slot_val[sb + base + q] = st_val[q]; }
        if (k == 0) counts[r] = total;
        if (n_mine > kStash || (k == 0 && total > K)) atomicOr(overflow, 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the next group's LDS writes stay behind this group's reads
    }
}

// The candidate walk with the exact plateau machine run only where it can matter.  Every candidate scipy keeps passes
// `height`: its plateau value is >= hmin, so its plateau STARTS at a detection value >= hmin.  A wave reads a group of
// consecutive uniform records (16 KiB, 16-byte pieces, every byte of the pool once) and tests each piece in registers, in
// float32 (two or three instructions per sample), for a value that may reach hmin -- a test that can only err towards
// "hot" (launch_find_peaks_hot derives the float32 bound h32).  The float64 machine of k_find_peaks_staged then runs with a
// lane per HOT piece, reading its few samples back through L2: it finds the candidates whose plateau starts in its piece,
// exactly as a lane of the staged kernel does for its stretch.  With the reference's default height (30 on the derivative of
// a filtered waveform) one piece in ~200 is hot.  Same slots, counts and overflow flag as the staged kernel.
struct HotPeakArgs {
    int64_t off0;   // first sample of record 0 in the pool
    int32_t L;      // samples per record (multiple of 8)
    int32_t RW;     // records per wave
    float h32;      // float32 bound of `height`
    float inv_ppr;  // 1 / (pieces per record)
};
constexpr int kHotWaves = 4;  // waves per block, each with its own group

template <int SRC>
__global__ __launch_bounds__(kHotWaves * kWave) void k_find_peaks_hot(PoolView pool, RecView rec, PeakParams pp, HotPeakArgs ha,
                                                                      int K, int32_t* __restrict__ counts,
                                                                      int32_t* __restrict__ slot_pos,
                                                                      double* __restrict__ slot_val, int* __restrict__ overflow) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    constexpr int ES = SRC == WFA_SRC_RAW ? 2 : 4;
    constexpr int PIECE = 16 / ES;  // samples per 16-byte piece
    constexpr int kRounds = 16;     // pieces per lane: 1024 per wave
    __shared__ uint16_t s_hot_all[kHotWaves][kRounds * kWave];
    struct RecKey { float b32; uint32_t sgn; };  // float32 baseline, sign bit of "signal = -(w - baseline)"
    __shared__ double s_bl_all[kHotWaves][2 * kWave];  // (a lane past the group's last piece reads up to 63 records further)
    __shared__ RecKey s_key_all[kHotWaves][2 * kWave];
    __shared__ int s_cnt_all[kHotWaves][kWave];
    const int lane = lane_id(), wave = wave_in_block();
    uint16_t* s_hot = s_hot_all[wave];
    double* s_bl = s_bl_all[wave];
    RecKey* s_key = s_key_all[wave];
    int* s_cnt = s_cnt_all[wave];
    const int L = ha.L, RW = ha.RW;
    const int ppr = L / PIECE;  // pieces per record
    const int64_t r0 = ((int64_t)blockIdx.x * kHotWaves + wave) * RW;
    if (r0 >= rec.R) return;  // (no block-wide barrier below: the waves are independent)
    const int nrec = (int)(rec.R - r0 < RW ? rec.R - r0 : RW);
    const int pieces = nrec * ppr;  // <= 1024
    u4 pf[kRounds];
    {
        const u4* __restrict__ src = reinterpret_cast<const u4*>(SRC == WFA_SRC_RAW ? (const void*)pool.u16 : (const void*)pool.f32) +
                                     (ha.off0 * ES >> 4) + r0 * ppr;
#pragma unroll
        for (int t = 0; t < kRounds; ++t) {
            const int c = t * kWave + lane;
            pf[t] = src[c < pieces ? c : 0];
        }
        {
            const int64_t q = lane < nrec ? r0 + lane : r0;
            const double bl = rec.baseline[q];
            const uint32_t sgn = rec.pol[q] == WFA_POL_POSITIVE ? 0u : 0x80000000u;
            s_bl[lane] = bl; s_bl[kWave + lane] = bl;
            s_key[lane] = RecKey{(float)bl, sgn}; s_key[kWave + lane] = RecKey{(float)bl, sgn};
            s_cnt[lane] = 0;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int deriv = pp.use_derivative;
    const int n = deriv ? L - 1 : L;
    if (n < 3) {
        if (lane < nrec) counts[r0 + lane] = 0;
        return;
    }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint16_t* gu = pool.u16 ? pool.u16 + ha.off0 + r0 * L : nullptr;
    const float* gf = pool.f32 ? pool.f32 + ha.off0 + r0 * L : nullptr;
    auto sample = [&](int t, int j) -> float {  // sample j of this lane's piece of round t
        if constexpr (SRC == WFA_SRC_RAW) return (float)((pf[t][j >> 1] >> (16 * (j & 1))) & 0xffffu);
        else return __uint_as_float(pf[t][j]);
    };

    // make_hot(g)(w0, w1): may the detection value of samples (w0, w1) of record g be >= hmin?   make_det(g)(w0, w1): the value.
    auto run = [&](auto make_hot, auto make_det, auto deriv_tag) __attribute__((always_inline)) {
        constexpr int DERIV = decltype(deriv_tag)::value;
        // ---- pass 1: hot pieces, in (record, position) order.  (g, a) = record and first sample of the lane's piece,
        // advanced by 64 pieces per round; no branches: this loop is the kernel's instruction count
        int n_hot = 0;
        int g = (int)(((float)lane + 0.5f) * ha.inv_ppr);
        int a = (lane - g * ppr) * PIECE;
        const int wraps = (kWave + ppr - 1) / ppr;  // records a lane can cross per round
#pragma unroll
        for (int t = 0; t < kRounds; ++t) {
            if (t * kWave >= pieces) break;
            float w[PIECE + 1];
#pragma unroll
            for (int j = 0; j < PIECE; ++j) w[j] = sample(t, j);
            if constexpr (DERIV) {
                // the sample behind the piece: the next lane's first one (lane 63: lane 0 of the next round); the last piece
                // of a record does not use it (its last sample is no detection index of the derivative)
                const uint32_t wrap = (uint32_t)__builtin_amdgcn_readlane(__builtin_bit_cast(int, sample(t + 1 < kRounds ? t + 1 : t, 0)), 0);
                w[PIECE] = __uint_as_float(dpp_from_next_lane(wrap, __float_as_uint(w[0])));
            } else {
                w[PIECE] = 0.f;
            }
            const auto hot32 = make_hot(g);
            bool hot = false;
#pragma unroll
            for (int j = 0; j < PIECE - 1; ++j) hot |= hot32(w[j], w[j + 1]);
            hot |= hot32(w[PIECE - 1], w[PIECE]) & (!DERIV || a + PIECE < L);
            hot &= t * kWave + lane < pieces;
            const uint64_t m = __ballot(hot);
            if (m != 0) {
                if (hot) s_hot[n_hot + __popcll(m & lt_mask)] = (uint16_t)(t * kWave + lane);
                n_hot += __popcll(m);
            }
            a += kWave * PIECE;
            for (int q = 0; q < wraps; ++q) {
                const bool over = a >= L;
                a -= over ? L : 0;
                g += over ? 1 : 0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- pass 2: the plateau machine, a lane per hot piece (samples through L2: the wave has just read them)
        for (int e0 = 0; e0 < n_hot; e0 += kWave) {
            const bool act = e0 + lane < n_hot;
            const int c = act ? (int)s_hot[e0 + lane] : 0;
            const int g = (int)(((float)c + 0.5f) * ha.inv_ppr);
            const int a = act ? (c - g * ppr) * PIECE : 0;
            const int b = act ? (a + PIECE < n ? a + PIECE : n) : 0;
            const int gb = g * L;
            auto wv = [&](int m) { return SRC == WFA_SRC_RAW ? (float)gu[gb + m] : gf[gb + m]; };
            const auto det2 = make_det(g);
            constexpr int kStash = 4;  // plateau starts are two samples apart: at most four in a piece
            int n_mine = 0;
            int st_pos[kStash] = {0, 0, 0, 0};
            double st_val[kStash] = {0.0, 0.0, 0.0, 0.0};
            auto on_peak = [&](int c_start, int i, double c_val) {  // plateau [c_start, i - 1] ended by a fall at i
                const int peak = (c_start + i - 1) / 2;
                bool keep = c_val >= pp.hmin;
                if (keep && pp.has_threshold) {
                    const double lt = c_val - det2(wv(peak - 1), DERIV ? wv(peak) : 0.f);
                    const double rt = c_val - det2(wv(peak + 1), DERIV ? wv(peak + 2 < L ? peak + 2 : L - 1) : 0.f);
                    keep = (lt < rt ? lt : rt) >= pp.tmin;
                }
                if (keep) {
#pragma unroll
                    for (int q = 0; q < kStash; ++q)
                        if (n_mine == q) { st_pos[q] = peak; st_val[q] = c_val; }
                    ++n_mine;
                }
            };
            bool have = false;
            int c_start = 0;
            double c_val = 0.0;
            // samples a - 1 .. a + PIECE of the record (clamped), all requested before the first is used
            float ws[PIECE + 2];
#pragma unroll
            for (int j = 0; j < PIECE + 2; ++j) {
                const int m = a - 1 + j;
                ws[j] = wv(m < 0 ? 0 : (m < L ? m : L - 1));
            }
            double x_prev = a > 0 ? det2(ws[0], DERIV ? ws[1] : 0.f) : 0.0;
#pragma unroll
            for (int u = 0; u < PIECE; ++u) {
                const int i = a + u;
                const double x = det2(ws[u + 1], DERIV ? ws[u + 2] : 0.f);
                const bool in = i < b;
                const bool actv = in && i > 0;
                const bool fall = actv && have && x < c_val;
                if (fall) on_peak(c_start, i, c_val);
                const bool set = actv && (have ? x > c_val : (x_prev < x && i < n - 1));
                have = (have && !fall) || (actv && set);
                c_start = set ? i : c_start;
                c_val = set ? x : c_val;
                x_prev = in ? x : x_prev;
            }
            for (int i = b; __ballot(have && i < n) != 0; ++i) {  // an open candidate is followed to its end
                const bool go = have && i < n;
                const int m = go ? i : 0;
                const double x = det2(wv(m), DERIV ? wv(m + 1 < L ? m + 1 : L - 1) : 0.f);
                if (go && x < c_val) on_peak(c_start, i, c_val);
                have = go && x == c_val;  // a fall ends it as a peak, a rise as the start of a later piece's plateau
            }
            // slots: candidates of a record in position order = in lane order, continuing the record's earlier batches
            const int gk = act ? g : kWave;
            int tot;
            const int excl = wave_excl_scan_i32(n_mine, tot);
            const int gprev = __shfl_up(gk, 1, kWave);
            const int gnext = __shfl_down(gk, 1, kWave);
            const uint64_t heads = __ballot(lane == 0 || gk != gprev);
            const int hl = 63 - __builtin_clzll(heads & (lt_mask | (1ull << lane)));
            const int slot0 = (act ? s_cnt[g] : 0) + excl - __shfl(excl, hl, kWave);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (act) {
                const int64_t sb = (r0 + g) * K;
#pragma unroll
                for (int q = 0; q < kStash; ++q)
                    if (n_mine > q && slot0 + q < K) { slot_pos[sb + slot0 + q] = st_pos[q]; slot_val[sb + slot0 + q] = st_val[q]; }
                if (lane == kWave - 1 || gk != gnext) s_cnt[g] = slot0 + n_mine;
                if (n_mine > kStash) atomicOr(overflow, 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    };
    {
        const float h32 = ha.h32;
        const double hmin = pp.hmin;
        auto rec32 = [](float w, float b32, uint32_t sgn) { return __uint_as_float(__float_as_uint(w - b32) ^ sgn); };
        if (!pp.rows) {
            if (deriv)
                run([&](int g) { const RecKey k = s_key[g];
                                 return [=](float w0, float w1) { return rec32(w1, k.b32, k.sgn) - rec32(w0, k.b32, k.sgn) >= h32; }; },
                    [&](int g) { const RecKey k = s_key[g];
                                 return [=](float w0, float w1) { return (double)rec32(w1, k.b32, k.sgn) - (double)rec32(w0, k.b32, k.sgn); }; },
                    std::integral_constant<int, 1>{});
            else
                run([&](int g) { const RecKey k = s_key[g];
                                 return [=](float w0, float) { return rec32(w0, k.b32, k.sgn) >= h32; }; },
                    [&](int g) { const RecKey k = s_key[g];
                                 return [=](float w0, float) { return (double)rec32(w0, k.b32, k.sgn) - 0.0; }; },
                    std::integral_constant<int, 0>{});
        } else if (!deriv) {
            run([&](int g) { const double b64 = s_bl[g]; return [=](float w0, float) { return b64 - (double)w0 >= hmin; }; },
                [&](int g) { const double b64 = s_bl[g]; return [=](float w0, float) { return b64 - (double)w0; }; },
                std::integral_constant<int, 0>{});
        } else if (SRC == WFA_SRC_RAW || pp.rows == WFA_PEAK_SIGNAL_ROWS_F64) {
            run([&](int) { return [=](float w0, float w1) { return -(w1 - w0) >= h32; }; },
                [&](int) { return [=](float w0, float w1) { return -((double)w1 - (double)w0); }; },
                std::integral_constant<int, 1>{});
        } else {
            run([&](int) { return [=](float w0, float w1) { return -(w1 - w0) >= h32; }; },
                [&](int) { return [=](float w0, float w1) { return (double)(-(w1 - w0)); }; },
                std::integral_constant<int, 1>{});
        }
    }
    if (lane < nrec) {
        const int tot = s_cnt[lane];
        counts[r0 + lane] = tot;
        if (tot > K) atomicOr(overflow, 1);
    }
}

__global__ __launch_bounds__(kPeakBlock) void k_peak_compact(int64_t R, int K, const int32_t* __restrict__ counts,
                                                             const int64_t* __restrict__ cand_start,
                                                             const int32_t* __restrict__ slot_pos,
                                                             const double* __restrict__ slot_val,
                                                             int32_t* __restrict__ cand_pos, double* __restrict__ cand_val,
                                                             int64_t* __restrict__ cand_rec) {
    const int64_t t = (int64_t)blockIdx.x * kPeakBlock + threadIdx.x;
    const int64_t r = t / K;
    const int k = (int)(t - r * K);
    if (r >= R || k >= counts[r]) return;
    const int64_t dst = cand_start[r] + k;
    cand_pos[dst] = slot_pos[t];
    cand_val[dst] = slot_val[t];
    cand_rec[dst] = r;
}

// _select_by_peak_distance on the candidate list of each record: visit candidates by descending value (ties: the
// later candidate first == a stable ascending argsort read backwards) and drop the not-yet-dropped neighbours
// closer than `distance`.  state: 1 = kept & unvisited, 2 = kept & visited, 0 = dropped.
__global__ __launch_bounds__(kPeakBlock) void k_peak_select(int64_t R, const int32_t* __restrict__ counts,
                                                            const int64_t* __restrict__ cand_start,
                                                            const int32_t* __restrict__ cand_pos,
                                                            const double* __restrict__ cand_val,
                                                            uint8_t* __restrict__ state, int distance) {
    const int64_t r = (int64_t)blockIdx.x * kPeakBlock + threadIdx.x;
    if (r >= R) return;
    const int K = counts[r];
    const int64_t b = cand_start[r];
    for (int k = 0; k < K; ++k) state[b + k] = 1;
    for (;;) {
        int best = -1;
        double bv = 0.0;
        for (int k = 0; k < K; ++k) {
            if (state[b + k] != 1) continue;
            const double v = cand_val[b + k];
            if (best < 0 || v >= bv) { best = k; bv = v; }
        }
        if (best < 0) break;
        state[b + best] = 2;
        const int pj = cand_pos[b + best];
        for (int k = best - 1; k >= 0 && pj - cand_pos[b + k] < distance; --k) state[b + k] = 0;
        for (int k = best + 1; k < K && cand_pos[b + k] - pj < distance; ++k) state[b + k] = 0;
    }
}

// one lane per candidate: prominence + width -> accept flag and the two interpolated intersection points
template <int SRC>
__global__ __launch_bounds__(kPeakBlock) void k_peak_eval(PoolView pool, RecView rec, PeakParams pp, int64_t n_cand,
                                                          const int64_t* __restrict__ cand_rec,
                                                          const int32_t* __restrict__ cand_pos,
                                                          const uint8_t* __restrict__ state,
                                                          int32_t* __restrict__ accept, double* __restrict__ ips) {
    // (Handing the lanes of a wave candidates of similar height -- a radix sort by value in front -- did not pay:
    // 1.58 ms against 1.21 ms, the walks of neighbouring candidates share cache lines.  Nor did a wave-level form of the
    // walks without a branch per sample -- both directions in one loop, every sample slot evaluated by every lane through
    // selects: identical rows, 0.92 ms against 0.70: 201 registers, 2 waves per SIMD, and the launch is a chain of load
    // round trips per wave, not instruction issue.)
    const int64_t k = (int64_t)blockIdx.x * kPeakBlock + threadIdx.x;
    if (k >= n_cand) return;
    int ok = 0;
    double l_ip = 0.0, r_ip = 0.0;
    if (!state || state[k]) {
        auto run = [&](auto mode_tag) __attribute__((always_inline)) {
            SignalAt<SRC, decltype(mode_tag)::value> S;
            S.bind(pool, rec, cand_rec[k], pp);
            ok = peak_passes<SRC>(S, cand_pos[k], pp, l_ip, r_ip) ? 1 : 0;
        };
        switch (signal_mode<SRC>(pp.rows, pp.use_derivative)) {  // wave-uniform: one form of det_of per launch
            case 0: run(std::integral_constant<int, 0>{}); break;
            case 1: run(std::integral_constant<int, 1>{}); break;
            case 2: run(std::integral_constant<int, 2>{}); break;
            case 3: run(std::integral_constant<int, 3>{}); break;
            default: run(std::integral_constant<int, 4>{}); break;
        }
    }
    accept[k] = ok;
    ips[2 * k] = l_ip;
    ips[2 * k + 1] = r_ip;
}

// one lane per candidate: accepted ones write their HIT_DTYPE row at the scanned offset (candidate order is
// (record, position) order, so the rows are too)
template <int SRC>
__global__ __launch_bounds__(kPeakBlock) void k_peak_rows(PoolView pool, RecView rec, PeakParams pp, int64_t n_cand,
                                                          const int64_t* __restrict__ cand_rec,
                                                          const int32_t* __restrict__ cand_pos,
                                                          const int32_t* __restrict__ accept,
                                                          const int64_t* __restrict__ row_start,
                                                          const double* __restrict__ ips, uint8_t* __restrict__ out,
                                                          int* __restrict__ err) {
    __shared__ double s_pw[kPairwiseLevels][kPeakBlock];  // 'diff' height: numpy's pairwise levels
    const int64_t k = (int64_t)blockIdx.x * kPeakBlock + threadIdx.x;
    if (k >= n_cand || !accept[k]) return;
    const int64_t r = cand_rec[k];
    SignalAt<SRC> S;
    S.bind(pool, rec, r, pp);
    write_peak_row(S, rec, r, cand_pos[k], ips[2 * k], ips[2 * k + 1], pp,
                   reinterpret_cast<uint32_t*>(out + row_start[k] * 48), err, &s_pw[0][threadIdx.x]);
}

// =============================================================================================
// K3: Butterworth band-pass, scipy.signal.sosfiltfilt  (filtering.py:84-101,198-224)
// =============================================================================================
// sosfiltfilt is a recursive filter: strictly sequential along a record, independent across records,
// so one lane owns one record and executes scipy's loop literally (float64, no fused multiply-add):
//   ext  = odd extension of the float32 wave by `edge` samples at both ends (float32 arithmetic),
//   fwd  = sosfilt(sos, ext, zi = zi0 * ext[0])       direct form II transposed, section by section:
//              x_new = b0*x + z0 ; z0 = b1*x - a1*x_new + z1 ; z1 = b2*x - a2*x_new ; x = x_new
//   bwd  = sosfilt(sos, reverse(fwd), zi = zi0 * fwd[-1]) ; result = reverse(bwd)[edge:-edge] -> float32
// The forward output lives in a float64 scratch laid out [sample][record-in-batch] so the 64 lanes of a
// wave read and write consecutive addresses.  Records with L <= padlen are copied (filtering.py:221-222).
constexpr int kMaxSections = 8;

struct SosParams {
    int n_sections;
    int edge;                       // padlen
    double sos[kMaxSections][6];
    double zi[kMaxSections][2];
};

__global__ __launch_bounds__(kBlock) void k_sosfiltfilt(PoolView pool, RecView rec, SosParams sp, int64_t r_begin,
                                                        int64_t r_end, double* __restrict__ scratch,
                                                        int64_t batch_stride, float* __restrict__ out) {
    const int64_t r = r_begin + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= r_end) return;
    const int L = rec.len[r];
    if (L <= 0) return;
    const int64_t off = rec.off[r];
    const uint16_t* __restrict__ x = pool.u16 + off;
    float* __restrict__ y = out + off;
    const int edge = sp.edge;
    if (L <= edge) {
        for (int i = 0; i < L; ++i) y[i] = (float)x[i];
        return;
    }
    double* __restrict__ col = scratch + (r - r_begin);  // element n at col[n * batch_stride]
    const int n_ext = L + 2 * edge;
    // float32 odd extension (scipy _arraytools.odd_ext on the float32 wave)
    auto ext_at = [&](int n) -> double {
        if (n < edge) return (double)(2.0f * (float)x[0] - (float)x[edge - n]);
        if (n < edge + L) return (double)(float)x[n - edge];
        return (double)(2.0f * (float)x[L - 1] - (float)x[L - 2 - (n - edge - L)]);
    };
    double z0[kMaxSections], z1[kMaxSections];
    const double x0 = ext_at(0);
#pragma unroll
    for (int s = 0; s < kMaxSections; ++s) {
        z0[s] = s < sp.n_sections ? sp.zi[s][0] * x0 : 0.0;
        z1[s] = s < sp.n_sections ? sp.zi[s][1] * x0 : 0.0;
    }
    double last = 0.0;
    for (int n = 0; n < n_ext; ++n) {
        double xc = ext_at(n);
#pragma unroll
        for (int s = 0; s < kMaxSections; ++s) {
            if (s < sp.n_sections) {
                const double xn = sp.sos[s][0] * xc + z0[s];
                z0[s] = sp.sos[s][1] * xc - sp.sos[s][4] * xn + z1[s];
                z1[s] = sp.sos[s][2] * xc - sp.sos[s][5] * xn;
                xc = xn;
            }
        }
        col[(int64_t)n * batch_stride] = xc;
        last = xc;
    }
#pragma unroll
    for (int s = 0; s < kMaxSections; ++s) {
        z0[s] = s < sp.n_sections ? sp.zi[s][0] * last : 0.0;
        z1[s] = s < sp.n_sections ? sp.zi[s][1] * last : 0.0;
    }
    for (int n = n_ext - 1; n >= 0; --n) {
        double xc = col[(int64_t)n * batch_stride];
#pragma unroll
        for (int s = 0; s < kMaxSections; ++s) {
            if (s < sp.n_sections) {
                const double xn = sp.sos[s][0] * xc + z0[s];
                z0[s] = sp.sos[s][1] * xc - sp.sos[s][4] * xn + z1[s];
                z1[s] = sp.sos[s][2] * xc - sp.sos[s][5] * xn;
                xc = xn;
            }
        }
        const int i = n - edge;
        if (i >= 0 && i < L) y[i] = (float)xc;
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
    (void)sg;
    if (rec.R == 0) return hipSuccess;
    const unsigned grid = (unsigned)((rec.R + kFeatBlock - 1) / kFeatBlock);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_basic_features<WFA_SRC_RAW>), dim3(grid), dim3(kFeatBlock), 0, st, pool, rec, fp, out);
    else if (source == WFA_SRC_F32)
        hipLaunchKernelGGL((k_basic_features<WFA_SRC_F32>), dim3(grid), dim3(kFeatBlock), 0, st, pool, rec, fp, out);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_width_integral(hipStream_t st, int source, const PoolView& pool, const RecView& rec,
                                 const SgParams& sg, const WidthParams& wp, uint8_t* out) {
    (void)sg;
    if (rec.R == 0) return hipSuccess;
    const unsigned grid = (unsigned)((rec.R + kFeatBlock - 1) / kFeatBlock);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_width_integral<WFA_SRC_RAW>), dim3(grid), dim3(kFeatBlock), 0, st, pool, rec, wp, out);
    else if (source == WFA_SRC_F32)
        hipLaunchKernelGGL((k_width_integral<WFA_SRC_F32>), dim3(grid), dim3(kFeatBlock), 0, st, pool, rec, wp, out);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_sosfiltfilt(hipStream_t st, const PoolView& pool, const RecView& rec, int n_sections,
                              const double* sos, const double* zi, int edge, int64_t r_begin, int64_t r_end,
                              double* scratch, int64_t batch_stride, float* out) {
    SosParams sp{};
    sp.n_sections = n_sections;
    sp.edge = edge;
    for (int s = 0; s < n_sections; ++s) {
        for (int k = 0; k < 6; ++k) sp.sos[s][k] = sos[s * 6 + k];
        sp.zi[s][0] = zi[s * 2];
        sp.zi[s][1] = zi[s * 2 + 1];
    }
    const int64_t n = r_end - r_begin;
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_sosfiltfilt, dim3(grid), dim3(kBlock), 0, st, pool, rec, sp, r_begin, r_end, scratch,
                       batch_stride, out);
    return hipGetLastError();
}

hipError_t launch_find_hits_legacy(hipStream_t st, int source, bool fill, const PoolView& pool, int64_t n_rows, int32_t L,
                                   const double* baselines, double threshold, int32_t* counts, const int64_t* out_start,
                                   int64_t* out_event, int64_t* out_time) {
    if (n_rows == 0) return hipSuccess;
    if (source != WFA_SRC_RAW && source != WFA_SRC_F32) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_rows + kWavesPerBlock - 1) / kWavesPerBlock);
#define WFA_FH(SRC, F) \
    hipLaunchKernelGGL((k_find_hits_legacy<SRC, F>), dim3(grid), dim3(kBlock), 0, st, pool, n_rows, L, baselines, threshold, counts, out_start, out_event, out_time)
    if (source == WFA_SRC_RAW) { if (fill) WFA_FH(WFA_SRC_RAW, true); else WFA_FH(WFA_SRC_RAW, false); }
    else { if (fill) WFA_FH(WFA_SRC_F32, true); else WFA_FH(WFA_SRC_F32, false); }
#undef WFA_FH
    return hipGetLastError();
}

hipError_t launch_waveform_width(hipStream_t st, int source, const PoolView& pool, int64_t n_hits,
                                 const int64_t* position, const int64_t* row_index, int64_t n_rows, int32_t L,
                                 double rise_low, double rise_high, double fall_high, double fall_low,
                                 double sampling_rate, int interpolation, uint8_t* out, uint8_t* valid) {
    if (n_hits == 0) return hipSuccess;
    WidthHitParams wp{rise_low, rise_high, fall_high, fall_low, sampling_rate, interpolation, L, n_rows};
    const unsigned grid = (unsigned)((n_hits + 127) / 128);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_waveform_width<WFA_SRC_RAW>), dim3(grid), dim3(128), 0, st, pool, n_hits, position, row_index, wp, out, valid);
    else if (source == WFA_SRC_F32)
        hipLaunchKernelGGL((k_waveform_width<WFA_SRC_F32>), dim3(grid), dim3(128), 0, st, pool, n_hits, position, row_index, wp, out, valid);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_find_peaks(hipStream_t st, int source, bool fill, const PoolView& pool, const RecView& rec,
                             const PeakParams& pp, int32_t* counts, const int64_t* out_start, int32_t* cand_pos,
                             double* cand_val, int64_t* cand_rec) {
    if (rec.R == 0) return hipSuccess;
    if (source != WFA_SRC_RAW && source != WFA_SRC_F32) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((rec.R + kPeakBlock - 1) / kPeakBlock);
#define WFA_PK(SRC, F) \
    hipLaunchKernelGGL((k_find_peaks<SRC, F>), dim3(grid), dim3(kPeakBlock), 0, st, pool, rec, pp, counts, out_start, cand_pos, cand_val, cand_rec)
    if (source == WFA_SRC_RAW) { if (fill) WFA_PK(WFA_SRC_RAW, true); else WFA_PK(WFA_SRC_RAW, false); }
    else { if (fill) WFA_PK(WFA_SRC_F32, true); else WFA_PK(WFA_SRC_F32, false); }
#undef WFA_PK
    return hipGetLastError();
}

hipError_t launch_find_peaks_slots(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                                   int K, int32_t* counts, int32_t* slot_pos, double* slot_val, int* overflow) {
    if (rec.R == 0) return hipSuccess;
    const unsigned grid = (unsigned)((rec.R + kPeakBlock - 1) / kPeakBlock);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_find_peaks_slots<WFA_SRC_RAW>), dim3(grid), dim3(kPeakBlock), 0, st, pool, rec, pp, K, counts, slot_pos, slot_val, overflow);
    else
        hipLaunchKernelGGL((k_find_peaks_slots<WFA_SRC_F32>), dim3(grid), dim3(kPeakBlock), 0, st, pool, rec, pp, K, counts, slot_pos, slot_val, overflow);
    return hipGetLastError();
}

// false: the layout is not covered (the lane-per-record walk runs instead)
bool launch_find_peaks_staged(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                              int64_t off0, int L, int K, int32_t* counts, int32_t* slot_pos, double* slot_val, int* overflow,
                              hipError_t* err) {
    *err = hipSuccess;
    if (rec.R == 0 || L < 8 || (L & 7) || (off0 & 7)) return false;
    if (source != WFA_SRC_RAW && source != WFA_SRC_F32) return false;
    const int cap = source == WFA_SRC_RAW ? 8192 : 4096;  // samples staged per wave: 16 KiB of LDS
    int sh = 0;
    while (sh < 6 && (int64_t)(kWave >> sh) * L > cap) ++sh;
    if ((int64_t)(kWave >> sh) * L > cap) return false;
    while (sh < 6 && (L >> sh) > 256) ++sh;  // at most 256 samples per lane
    StagedPeakArgs sa{off0, L, sh};
    const int RW = kWave >> sh;
    const size_t lds = (size_t)RW * L * (source == WFA_SRC_RAW ? 2 : 4) + 16;
    const int64_t n_groups = (rec.R + RW - 1) / RW;
    // one group per wave: a resident-set grid of persistent waves (each with its next group in flight) measured 2.38 ms
    // against 1.56 ms -- the walks of a group end at very different times (open candidates are followed past the lane's
    // stretch), and many small blocks balance that better than the prefetch hides latency
    const unsigned grid = (unsigned)n_groups;
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_find_peaks_staged<WFA_SRC_RAW>), dim3(grid), dim3(kWave), lds, st, pool, rec, pp, sa, K, counts, slot_pos, slot_val, overflow);
    else
        hipLaunchKernelGGL((k_find_peaks_staged<WFA_SRC_F32>), dim3(grid), dim3(kWave), lds, st, pool, rec, pp, sa, K, counts, slot_pos, slot_val, overflow);
    *err = hipGetLastError();
    return true;
}

// false: the layout or the parameters are not covered (the staged walk runs instead)
bool launch_find_peaks_hot(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                           int64_t off0, int L, int K, int32_t* counts, int32_t* slot_pos, double* slot_val, int* overflow,
                           hipError_t* err) {
    *err = hipSuccess;
    if (rec.R == 0 || L < 8 || (L & 7) || (off0 & 7)) return false;
    if (source != WFA_SRC_RAW && source != WFA_SRC_F32) return false;
    if (!std::isfinite(pp.hmin)) return false;  // no height bound: every piece would be hot
    const int cap = source == WFA_SRC_RAW ? 8192 : 4096;  // samples per wave: 1024 pieces of 16 bytes
    if (L > cap) return false;
    const int RW = cap / L < kWave ? cap / L : kWave;
    // float32 bound of `height`.  A detection value is x = fl64(d) for a real d (a difference of float32 samples, or one
    // of them); the first pass sees k = fl32(d).  Rounding is monotonic: x >= hmin gives fl32(x) >= fl32(hmin), and fl32(d)
    // is fl32(x) or its neighbour below (d and x differ by half a float64 ulp at most).  Two float32 steps below
    // fl32(hmin) is therefore a bound no value that passes `height` can fall under.
    float h32 = (float)pp.hmin;
    h32 = std::nextafterf(std::nextafterf(h32, -INFINITY), -INFINITY);
    const int piece = source == WFA_SRC_RAW ? 8 : 4;
    HotPeakArgs ha{off0, L, RW, h32, 1.0f / (float)(L / piece)};
    const int64_t groups = (rec.R + RW - 1) / RW;
    const unsigned grid = (unsigned)((groups + kHotWaves - 1) / kHotWaves);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_find_peaks_hot<WFA_SRC_RAW>), dim3(grid), dim3(kHotWaves * kWave), 0, st, pool, rec, pp, ha, K, counts, slot_pos, slot_val, overflow);
    else
        hipLaunchKernelGGL((k_find_peaks_hot<WFA_SRC_F32>), dim3(grid), dim3(kHotWaves * kWave), 0, st, pool, rec, pp, ha, K, counts, slot_pos, slot_val, overflow);
    *err = hipGetLastError();
    return true;
}

hipError_t launch_peak_compact(hipStream_t st, int64_t R, int K, const int32_t* counts, const int64_t* cand_start,
                               const int32_t* slot_pos, const double* slot_val, int32_t* cand_pos, double* cand_val,
                               int64_t* cand_rec) {
    if (R == 0) return hipSuccess;
    const int64_t n = R * K;
    hipLaunchKernelGGL(k_peak_compact, dim3((unsigned)((n + kPeakBlock - 1) / kPeakBlock)), dim3(kPeakBlock), 0, st, R, K, counts,
                       cand_start, slot_pos, slot_val, cand_pos, cand_val, cand_rec);
    return hipGetLastError();
}

hipError_t launch_peak_select(hipStream_t st, int64_t R, const int32_t* counts, const int64_t* cand_start,
                              const int32_t* cand_pos, const double* cand_val, uint8_t* state, int distance) {
    if (R == 0) return hipSuccess;
    const unsigned grid = (unsigned)((R + kPeakBlock - 1) / kPeakBlock);
    hipLaunchKernelGGL(k_peak_select, dim3(grid), dim3(kPeakBlock), 0, st, R, counts, cand_start, cand_pos, cand_val,
                       state, distance);
    return hipGetLastError();
}

hipError_t launch_peak_eval(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                            int64_t n_cand, const int64_t* cand_rec, const int32_t* cand_pos, const uint8_t* state,
                            int32_t* accept, double* ips) {
    if (n_cand == 0) return hipSuccess;
    if (source != WFA_SRC_RAW && source != WFA_SRC_F32) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_cand + kPeakBlock - 1) / kPeakBlock);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_peak_eval<WFA_SRC_RAW>), dim3(grid), dim3(kPeakBlock), 0, st, pool, rec, pp, n_cand, cand_rec, cand_pos, state, accept, ips);
    else
        hipLaunchKernelGGL((k_peak_eval<WFA_SRC_F32>), dim3(grid), dim3(kPeakBlock), 0, st, pool, rec, pp, n_cand, cand_rec, cand_pos, state, accept, ips);
    return hipGetLastError();
}

hipError_t launch_peak_rows(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                            int64_t n_cand, const int64_t* cand_rec, const int32_t* cand_pos, const int32_t* accept,
                            const int64_t* row_start, const double* ips, uint8_t* out, int* err) {
    if (n_cand == 0) return hipSuccess;
    if (source != WFA_SRC_RAW && source != WFA_SRC_F32) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n_cand + kPeakBlock - 1) / kPeakBlock);
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_peak_rows<WFA_SRC_RAW>), dim3(grid), dim3(kPeakBlock), 0, st, pool, rec, pp, n_cand, cand_rec, cand_pos, accept, row_start, ips, out, err);
    else
        hipLaunchKernelGGL((k_peak_rows<WFA_SRC_F32>), dim3(grid), dim3(kPeakBlock), 0, st, pool, rec, pp, n_cand, cand_rec, cand_pos, accept, row_start, ips, out, err);
    return hipGetLastError();
}

int hit_runs_block() { return kRunsBlock; }

bool sg_mask_supported(const SgParams& sg) {
    return sg.int_ok && sg.W >= 5 && sg.W <= 15 && (sg.W & 1);
}

hipError_t launch_sg_mask(hipStream_t st, bool fused_baseline, int max_len, const PoolView& pool,
                          const RecView& rec, const SgParams& sg, const MaskParams& mp) {
    const int grid = grid_for_records(rec.R);
    const bool pf2 = (max_len + 7 + 511) / 512 <= 2;
#define WFA_MASK2(WW, FB, PFN) \
    hipLaunchKernelGGL((k_sg_mask<WW, FB, PFN>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, mp)
#define WFA_MASK(WW)                                                                            \
    case WW:                                                                                    \
        if (fused_baseline) { if (pf2) WFA_MASK2(WW, true, 2); else WFA_MASK2(WW, true, 4); }  \
        else { if (pf2) WFA_MASK2(WW, false, 2); else WFA_MASK2(WW, false, 4); }               \
        break;
    switch (sg.W) {
        WFA_MASK(5)
        WFA_MASK(7)
        WFA_MASK(9)
        WFA_MASK(11)
        WFA_MASK(13)
        WFA_MASK(15)
        default: return hipErrorInvalidValue;
    }
#undef WFA_MASK
#undef WFA_MASK2
    return hipGetLastError();
}

hipError_t launch_savgol_span(hipStream_t st, const PoolView& pool, const RecView& rec, const SgParams& sg,
                              const SpanParams& sp, float* out) {
    int64_t g = (sp.n_spans + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;
    const int grid = (int)g;
#define WFA_SVS(WW)                                                                                                  \
    case WW:                                                                                                         \
        if (sp.S > sp.L) hipLaunchKernelGGL((k_savgol_span<WW, true>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, sp, out); \
        else hipLaunchKernelGGL((k_savgol_span<WW, false>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, sp, out);   \
        break;
    switch (sg.W) {
        WFA_SVS(5)
        WFA_SVS(7)
        WFA_SVS(9)
        WFA_SVS(11)
        WFA_SVS(13)
        WFA_SVS(15)
        default: return hipErrorInvalidValue;
    }
#undef WFA_SVS
    return hipGetLastError();
}

// uniform records whose length is not a multiple of 16: the same kernel on the padded shadow layout (stride
// roundup16(L)).  The last chunk of a slot must hold the H right-edge samples: L % 16 >= H.
bool sg_mask_span16_padded_supported(const SgParams& sg, int32_t L) {
    return sg.int_ok && (L % 16) != 0 && (L % 16) >= sg.W / 2 && L >= 32 && sg.W >= 5 && sg.W <= 11;
}

// shadow layout: record r from src[off0 + r * L ...] to dst[r * S ...], padding zeroed; one thread per 8 samples
__global__ __launch_bounds__(kBlock) void k_pad_rows(const uint16_t* __restrict__ src, int64_t off0, int32_t L, int32_t S,
                                                     int64_t R, uint16_t* __restrict__ dst, int64_t* __restrict__ dst_off) {
    const int per_row = S >> 3;
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= R * per_row) return;
    const int64_t r = g / per_row;
    const int c = (int)(g - r * per_row);
    const int64_t idx = off0 + r * (int64_t)L + c * 8;
    const uint16_t* __restrict__ p = src + idx;
    uint32_t w[4];
    if (c * 8 + 8 <= L && (idx & 3) == 0) {  // whole chunk inside the record, source 8-byte aligned
        const uint2 a = *reinterpret_cast<const uint2*>(p), b = *reinterpret_cast<const uint2*>(p + 4);
        w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y;
    } else if (c * 8 + 8 <= L && (idx & 1) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const uint32_t*>(p + 2 * j);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = c * 8 + 2 * j;
            const uint32_t lo = i < L ? p[2 * j] : 0u, hi = i + 1 < L ? p[2 * j + 1] : 0u;
            w[j] = lo | (hi << 16);
        }
    }
    *reinterpret_cast<uint4*>(dst + r * (int64_t)S + c * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    if (c == 0) dst_off[r] = r * (int64_t)S;
}

hipError_t launch_pad_rows(hipStream_t st, const uint16_t* src, int64_t off0, int32_t L, int32_t S, int64_t R,
                           uint16_t* dst, int64_t* dst_off) {
    if (R == 0) return hipSuccess;
    const int64_t n = R * (S >> 3);
    hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, src, off0, L, S, R, dst, dst_off);
    return hipGetLastError();
}

bool sg_mask_span16_supported(const SgParams& sg, int L) {
    return sg.int_ok && (L % 16) == 0 && L >= 32 && sg.W >= 5 && sg.W <= 11;
}

hipError_t launch_sg_mask_span16(hipStream_t st, bool fused_baseline, const PoolView& pool, const RecView& rec,
                                 const SgParams& sg, const MaskParams& mp, const SpanParams& sp) {
    int64_t g = (sp.n_spans + kWavesPerBlock - 1) / kWavesPerBlock;
    // three rounds of the resident set: with exactly one round (persistent waves, spans dealt round-robin) the
    // 19 531 spans of the 10^9-sample chunk give 4.77 spans per wave, i.e. the last round runs 77 % full; shorter
    // blocks let the dispatcher even that out (grid x1 / x2 / x3 / x5 / x8: 0.695 / 0.677 / 0.662 / 0.677 / 0.678 ms)
    const int64_t resident = 3 * 256 * WFA_SPAN_WAVES;
    if (g < 1) g = 1;
    if (g > resident) g = resident;
    const int grid = (int)g;
#define WFA_SPAN16_LAUNCH(WW, FB, PD) \
    hipLaunchKernelGGL((k_sg_mask_span16<WW, FB, PD>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, mp, sp)
#define WFA_SPAN16(WW)                                                                                              \
    case WW:                                                                                                        \
        if (sp.S > sp.L) {                                                                                          \
            if (fused_baseline) WFA_SPAN16_LAUNCH(WW, true, true); else WFA_SPAN16_LAUNCH(WW, false, true);         \
        } else {                                                                                                    \
            if (fused_baseline) WFA_SPAN16_LAUNCH(WW, true, false); else WFA_SPAN16_LAUNCH(WW, false, false);       \
        }                                                                                                           \
        break;
    switch (sg.W) {
        WFA_SPAN16(5)
        WFA_SPAN16(7)
        WFA_SPAN16(9)
        WFA_SPAN16(11)
        default: return hipErrorInvalidValue;
    }
#undef WFA_SPAN16_LAUNCH
#undef WFA_SPAN16
    return hipGetLastError();
}

hipError_t launch_hit_runs(hipStream_t st, const RecView& rec, const uint8_t* bitmap, const int32_t* nhits,
                           const int64_t* out_start, int4* desc, const RowParams& rp) {
    if (rec.R == 0) return hipSuccess;
    const unsigned grid = (unsigned)((rec.R + kRunsBlock - 1) / kRunsBlock);
    hipLaunchKernelGGL(k_hit_runs, dim3(grid), dim3(kRunsBlock), (size_t)rp.stage_bytes, st, rec, bitmap, nhits, out_start, desc, rp);
    return hipGetLastError();
}

hipError_t launch_hit_rows_fast(hipStream_t st, const PoolView& pool, const RecView& rec, const SgParams& sg,
                                const RowParams& rp, int4* desc, int64_t n_hits, uint8_t* out, bool grouped) {
    if (n_hits == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_hits + kRowsHits - 1) / kRowsHits);
    const unsigned grid_flat = (unsigned)((n_hits + kFlatHits - 1) / kFlatHits);
#define WFA_ROWS(WW)                                                                                                              \
    case WW:                                                                                                                      \
        if (grouped) hipLaunchKernelGGL((k_hit_rows_grp<WW>), dim3(grid), dim3(kRowsBlock), 0, st, pool, rec, sg, rp, desc, n_hits, out); \
        else hipLaunchKernelGGL((k_hit_rows_flat<WW>), dim3(grid_flat), dim3(kWave), 0, st, pool, rec, sg, rp, desc, n_hits, out);       \
        break;
    switch (sg.W) {
        WFA_ROWS(5)
        WFA_ROWS(7)
        WFA_ROWS(9)
        WFA_ROWS(11)
        WFA_ROWS(13)
        WFA_ROWS(15)
        default: return hipErrorInvalidValue;
    }
#undef WFA_ROWS
    return hipGetLastError();
}

hipError_t launch_hit_rows_literal(hipStream_t st, int source, const PoolView& pool, const RecView& rec,
                                   const SgParams& sg, const RowParams& rp, const int4* desc, int64_t n_hits,
                                   bool only_flagged, uint8_t* out) {
    if (n_hits == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_hits + kBlock - 1) / kBlock);
    const int flag = only_flagged ? 1 : 0;
    if (source == WFA_SRC_RAW)
        hipLaunchKernelGGL((k_hit_rows_literal<WFA_SRC_RAW>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, rp, desc, n_hits, flag, out);
    else if (source == WFA_SRC_F32)
        hipLaunchKernelGGL((k_hit_rows_literal<WFA_SRC_F32>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, rp, desc, n_hits, flag, out);
    else
        hipLaunchKernelGGL((k_hit_rows_literal<WFA_SRC_SG_FUSED>), dim3(grid), dim3(kBlock), 0, st, pool, rec, sg, rp, desc, n_hits, flag, out);
    return hipGetLastError();
}

}  // namespace wfa
