// Per-record features on uniform records (BasicFeaturesPlugin records branch, cpu/basic_features.py:108-195;
// WaveformWidthIntegralPlugin, cpu/waveform_width_integral.py:83-231).
//
// The reference reduces with numpy, whose float64 results depend on the order of the additions; the rows must be bit
// identical.  The general kernels (wfa_kernels.hip: k_basic_features / k_width_integral) give every lane a whole record
// and walk it in numpy's order: correct for any layout, but a lane-per-record walk reads 16 bytes at a 1600-byte stride
// and runs 800 dependent float64 additions (0.12-0.15 of the HBM roofline).  Here numpy's order is mapped onto lanes:
//   np.sum  = pairwise_sum (umath/loops_utils.h.src): halves split at multiples of 8 down to leaves of <= 128 elements,
//             a leaf = 8 interleaved accumulators r_j = sum over m of x[a + 8 m + j], combined as
//             ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then its n % 8 tail added sequentially.
//             LANE = LEAF: an 800-sample record has exactly 8 leaves (96 + 104, four times), so a wave takes a group of
//             8 consecutive records, stages their 12.8 KB in LDS with coalesced 16-byte loads (the next group's loads
//             are in flight in registers meanwhile), and every lane walks its leaf 16 bytes at a time with the 8
//             accumulators in registers: the same additions on the same operands in the same order, 8 independent
//             chains per lane.  The leaves of a record then combine along numpy's recursion tree from a host-built plan
//             (PwPlan): lane butterflies when the tree is balanced, through LDS otherwise.
//   min / max / max|diff| are order independent and exact on the raw integers (two samples per packed operation).
//   np.cumsum + np.searchsorted: strictly sequential in numpy.  Each lane keeps the sequential sums of its own leaf
//             (one value per 8 samples), a scan over the record's lanes places them, and the crossing is located inside
//             one chunk.  Sums formed in another order may differ from numpy's in the last bits, so the INDEX is only
//             accepted when no value around the crossing lies within 8 L 2^-53 |target| of the target; the other records
//             (a few per thousand on integer data with a 1/40 baseline: exact ties are common) go to a list that
//             k_width_ties re-walks, one lane per record, in numpy's order.
#include <cstring>
#include <vector>

#include "wfa_common.hpp"
#include "wfa_device.hpp"
#include "wfa_kernels.hpp"

namespace wfa {

// numpy pairwise_sum of n <= 8192 elements: leaves in array order and the combine tree (slot = index of the leaf)
struct PwPlan {
    int16_t a[64], len[64];
    int8_t partner[8][64];  // level l: value[slot] += value[partner[l][slot]] (or -1)
    int32_t n_leaf, n_level, n, max_len, min_len;
    int32_t dpp_tree;  // the leaves pair up as a balanced binary tree of <= 16 leaves: lane butterflies combine them
    int32_t pad[2];
};

static int pw_build(PwPlan& p, int a, int n, int& depth_out) {
    if (n <= 128) {
        const int slot = p.n_leaf++;
        p.a[slot] = (int16_t)a;
        p.len[slot] = (int16_t)n;
        if (n > p.max_len) p.max_len = n;
        if (n < p.min_len) p.min_len = n;
        depth_out = 0;
        return slot;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    int dl = 0, dr = 0;
    const int l = pw_build(p, a, n2, dl);
    const int r = pw_build(p, a + n2, n - n2, dr);
    const int level = dl > dr ? dl : dr;
    p.partner[level][l] = (int8_t)r;
    if (level + 1 > p.n_level) p.n_level = level + 1;
    depth_out = level + 1;
    return l;
}

static bool pw_plan(PwPlan& p, int n) {
    memset(&p, 0, sizeof(p));
    memset(p.partner, -1, sizeof(p.partner));
    p.n = n;
    p.min_len = 1 << 20;
    if (n <= 0) { p.min_len = 0; return true; }
    if (n > 8192) return false;  // numpy's reduce walks longer arrays in 8192-element blocks: general kernel
    int d = 0;
    pw_build(p, 0, n, d);
    if (p.min_len < 8) p.min_len = 0;  // a single short leaf: sequential
    if (p.n_leaf > 64 || p.n_level > 8) return false;
    p.dpp_tree = p.n_leaf <= 16 && (p.n_leaf & (p.n_leaf - 1)) == 0 && (1 << p.n_level) == p.n_leaf;
    for (int lv = 0; lv < p.n_level && p.dpp_tree; ++lv)
        for (int k = 0; k < p.n_leaf; ++k) {
            const int want = (k % (2 << lv)) == 0 ? k + (1 << lv) : -1;
            if (p.partner[lv][k] != want) { p.dpp_tree = 0; break; }
        }
    return true;
}

namespace {

constexpr int kFwBlock = 64;   // one wave: LDS (one staged group per wave) is what limits the waves per CU
constexpr int kFwWaves = kFwBlock / kWave;
constexpr int kFwMaxChunks = 16;   // 16-byte chunks per lane: staged group (<= 8192 samples per wave), leaf (<= 128 samples)
constexpr int kFwGroupSamples = 8192;
constexpr int kFwCUs = 256;  // MI355X
#ifndef WFA_FW_OCC
#define WFA_FW_OCC 2
#endif

typedef unsigned short fw_us2 __attribute__((ext_vector_type(2)));
typedef uint32_t fw_u4 __attribute__((ext_vector_type(4)));

struct FwParams {
    const uint16_t* pool;
    int64_t off0;
    int32_t L;
    int32_t p0, p1, c0, c1;  // height range, area range (python slices already resolved against L)
    int32_t gl_shift;        // lanes per record = 1 << gl_shift; records per wave = 64 >> gl_shift
    const double* fixed_bl;
    double q_low, q_high, dt;
    int32_t* ties;  // [0] = count, then the records whose quantile positions numpy's own order has to decide
};

// the three ways the reference forms a sample's term (basic_features.py:150-175, waveform_width_integral.py:180-190)
struct TermKnown {  // signals = wave - baseline in float32; s = signals or -signals
    float b32; uint32_t sgn;
    __device__ __forceinline__ double operator()(uint32_t x) const {
        return (double)__uint_as_float(__float_as_uint((float)x - b32) ^ sgn);
    }
};
struct TermWave {  // float64: wave - baseline (s = 1, nb = -baseline) or baseline - wave (s = -1, nb = baseline); the
    double s, nb;  // fma is exact here (one rounding, of the same real number) and keeps the sign of an exact zero
    __device__ __forceinline__ double operator()(uint32_t x) const { return __builtin_fma(s, (double)x, nb); }
};
struct TermMixed {  // records of both kinds in one wave
    TermKnown k; TermWave w; bool known;
    __device__ __forceinline__ double operator()(uint32_t x) const { return known ? k(x) : w(x); }
};

template <bool CLIP, class F>
__device__ __forceinline__ void chunk_terms(const fw_u4& v, const F& f, double (&t)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t[2 * i] = f(v[i] & 0xffffu);
        t[2 * i + 1] = f(v[i] >> 16);
    }
    if (CLIP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_fmax(t[i], 0.0);  // a -0.0 term never changes a sum that starts at +0.0
    }
}

// One leaf of numpy's pairwise_sum per lane: q = the leaf's first chunk in LDS, len its length.  Returns the leaf's sum;
// with CUM also the running (sequential) sum after each chunk, cs[m], the partial tail counted as one more chunk.
// cmin / cmax: the chunk counts of the shortest / longest leaf of the plan (wave-uniform): chunks below cmin are read four
// at a time without a per-lane test, so their LDS reads are in flight together (a lane without a leaf reads leaf 0 and
// its result is dropped by the caller).
template <bool CLIP, bool CUM, class F>
__device__ __forceinline__ double leaf_sum_lane(const fw_u4* q, int len, int cmin, int cmax, const F& f,
                                                double (&cs)[kFwMaxChunks + 1]) {
    const int cnt = len >> 3, nt = len & 7;
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double tot = 0.0;
    auto take = [&](int m, const fw_u4& v) __attribute__((always_inline)) {
        double t[8];
        chunk_terms<CLIP>(v, f, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = m == 0 ? t[j] : r[j] + t[j];  // r[j] = a[j]; r[j] += a[i + j]
        if (CUM) {
            // the running sum only LOCATES the crossing (leaf_crossings accepts an index when nothing lies within
            // 8 L 2^-53 of the target, numpy's own order decides the rest): a chunk enters it as one value, three
            // dependent additions deep instead of eight
            tot += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
    };
    constexpr int B = 4;  // chunks read together (registers: the basic-features kernel runs at 3 waves per SIMD)
#pragma unroll
    for (int mb = 0; mb < kFwMaxChunks; mb += B) {
        if (mb + B <= cmin) {
            fw_u4 v[B];
#pragma unroll
            for (int i = 0; i < B; ++i) v[i] = q[mb + i];
#pragma unroll
            for (int i = 0; i < B; ++i) { take(mb + i, v[i]); if (CUM) cs[mb + i] = tot; }
        } else {
#pragma unroll
            for (int m = mb; m < mb + B; ++m) {
                if (m < cmax && m < cnt) take(m, q[m]);
                if (CUM) cs[m] = tot;
            }
        }
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    if (cnt == 0) res = 0.0;  // n < 8: res = 0.; res += a[i]
    if (nt) {
        double t[8];
        chunk_terms<CLIP>(q[cnt], f, t);  // the staging area is padded: reading past the leaf is harmless
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (j < nt) { res += t[j]; if (CUM) tot += t[j]; }
    }
    if (CUM) cs[kFwMaxChunks] = tot;
    return res;
}

// Both reductions of a record from ONE walk of the leaf (MODE 2): the unclipped terms into rb (basic_features' area), the
// same terms clipped at zero into the width kernel's accumulators and running sums.  Same additions, same operands, same
// order as the two separate walks -- the chunk is read and converted once.
template <class F>
__device__ __forceinline__ double leaf_sum_lane_both(const fw_u4* q, int len, int cmin, int cmax, const F& f,
                                                     double (&cs)[kFwMaxChunks + 1], double& res_plain) {
    const int cnt = len >> 3, nt = len & 7;
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double tot = 0.0;
    auto take = [&](int m, const fw_u4& v) __attribute__((always_inline)) {
        double t[8];
        chunk_terms<false>(v, f, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[j] = m == 0 ? t[j] : rb[j] + t[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = __builtin_fmax(t[j], 0.0);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = m == 0 ? t[j] : r[j] + t[j];
        tot += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));  // (see leaf_sum_lane)
    };
    constexpr int B = 2;  // (two accumulator sets live here: four chunks in flight spilled)
#pragma unroll
    for (int mb = 0; mb < kFwMaxChunks; mb += B) {
        if (mb + B <= cmin) {
            fw_u4 v[B];
#pragma unroll
            for (int i = 0; i < B; ++i) v[i] = q[mb + i];
#pragma unroll
            for (int i = 0; i < B; ++i) { take(mb + i, v[i]); cs[mb + i] = tot; }
        } else {
#pragma unroll
            for (int m = mb; m < mb + B; ++m) {
                if (m < cmax && m < cnt) take(m, q[m]);
                cs[m] = tot;
            }
        }
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    double resb = ((rb[0] + rb[1]) + (rb[2] + rb[3])) + ((rb[4] + rb[5]) + (rb[6] + rb[7]));
    if (cnt == 0) { res = 0.0; resb = 0.0; }
    if (nt) {
        double t[8];
        chunk_terms<false>(q[cnt], f, t);
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (j < nt) { resb += t[j]; const double tc = __builtin_fmax(t[j], 0.0); res += tc; tot += tc; }
    }
    cs[kFwMaxChunks] = tot;
    res_plain = resb;
    return res;
}

// First samples of this lane's leaf at which the cumulative sum reaches t[0] and t[1] (INT32_MAX: not in this leaf).
// excl = the cumulative value in front of the leaf.  near[i] = the decision was closer than tol[i] (the order of the
// additions could change it).
template <class F>
__device__ __forceinline__ void leaf_crossings(const fw_u4* q, int a, int len, const F& f, const double (&cs)[kFwMaxChunks + 1],
                                               double excl, const double (&t)[2], const double (&tol)[2], int (&found)[2],
                                               bool (&near)[2]) {
    const int cnt = len >> 3, nt = len & 7, nch = cnt + (nt ? 1 : 0);
    int mm[2] = {0, 0};
    double start[2] = {0.0, 0.0};
#pragma unroll
    for (int m = 0; m < kFwMaxChunks; ++m) {  // cs is non-decreasing: the chunks in front of each crossing
        const double e = excl + cs[m];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool below = m < cnt && e < t[i];
            mm[i] += below ? 1 : 0;
            start[i] = below ? cs[m] : start[i];
        }
    }
    const double e_end = excl + cs[kFwMaxChunks];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (nt && mm[i] == cnt && e_end < t[i]) { mm[i] = cnt + 1; start[i] = cs[kFwMaxChunks]; }  // the partial chunk
        found[i] = INT32_MAX;
        near[i] = false;
        if (mm[i] >= nch) {  // not in this leaf; its last value may still be too close to the target to call
            near[i] = nch > 0 && t[i] - (excl + start[i]) <= tol[i];
            mm[i] = -1;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (mm[i] >= 0) {
            double tv[8];
            chunk_terms<true>(q[mm[i]], f, tv);
            const int valid = mm[i] < cnt ? 8 : nt;
            double run = excl + start[i];
            bool open = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double before = run;
                run = j < valid ? run + tv[j] : run;
                if (j < valid && open && run >= t[i]) {
                    open = false;
                    found[i] = a + mm[i] * 8 + j;
                    near[i] = run - t[i] <= tol[i] || t[i] - before <= tol[i];
                }
            }
            if (open) near[i] = t[i] - run <= tol[i];  // ended just below the target
        }
    }
}

// min / max over the aligned group of gl (power of two) lanes; every lane gets the result.  Steps inside a row of 16
// lanes are DPP moves (no LDS round trip): quad swaps, then mirrors, which pair lanes that already agree per quad / half.
__device__ __forceinline__ int group_min_i32(int v, int gl) {
    int o;
    if (gl > 1) { o = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false); v = o < v ? o : v; }
    if (gl > 2) { o = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false); v = o < v ? o : v; }
    if (gl > 4) { o = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false); v = o < v ? o : v; }  // row_half_mirror
    if (gl > 8) { o = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false); v = o < v ? o : v; }  // row_mirror
    for (int m = 16; m < gl; m <<= 1) { o = __shfl_xor(v, m, kWave); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int group_max_i32(int v, int gl) { return ~group_min_i32(~v, gl); }

// double across lanes with DPP: sel 0/1/2 as dpp_f64, 3 = row_mirror
__device__ __forceinline__ double fw_dpp_f64(double v, int sel) {
    if (sel < 3) return dpp_f64(v, sel);
    const long long b = __double_as_longlong(v);
    int lo = (int)(uint32_t)(uint64_t)b, hi = (int)(uint32_t)((uint64_t)b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
// the value D lanes below in the same row of 16 lanes (0.0 where there is none)
template <int D>
__device__ __forceinline__ double fw_row_shr_f64(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(uint32_t)(uint64_t)b, hi = (int)(uint32_t)((uint64_t)b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + D, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + D, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}

struct FwMeta {  // per-record columns as loaded, fetched one group ahead together with the samples (nothing here may
    double bl, fb;  // be computed on before the next group's turn: that would wait for the whole prefetch)
    int32_t pol;
};
struct FwCold {  // columns that are only copied into the row: loaded at the end of a group's turn, stored with its row
    int64_t ts;
    uint32_t board, chan;
};

// MODE 0: BASIC_FEATURES_DTYPE rows (36 B); MODE 1: WAVEFORM_WIDTH_INTEGRAL_DTYPE rows (52 B); MODE 2: both from one
// staging of the group (BASELINE config 3 reads the pool once for the two feature tables: area range = whole record, no
// fixed baselines -- the plugins' defaults; `out` = the basic rows, `out2` = the width rows).
// A wave takes a group of 64 >> gl_shift consecutive records; lane = (record of the group, leaf of the reduction).
// PFN: 16-byte chunks a lane stages per group (13 covers 8 records of up to 832 samples).
template <int MODE, int PFN>
__global__ __launch_bounds__(kFwBlock, WFA_FW_OCC) void k_features_leaf(FwParams fw, RecView rec, const PwPlan* __restrict__ plan_g,
                                                            uint8_t* __restrict__ out, uint8_t* __restrict__ out2) {
    constexpr bool DO_B = MODE != 1, DO_W = MODE != 0;
    __shared__ PwPlan plan;
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(plan_g);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&plan);
        for (int k = threadIdx.x; k < (int)(sizeof(PwPlan) / 4); k += kFwBlock) dst[k] = src[k];
    }
    __syncthreads();
    const int lane = lane_id(), wv = wave_in_block();
    const int L = fw.L, CH = L >> 3;
    const int GL = 1 << fw.gl_shift, RW = kWave >> fw.gl_shift;
    const int g = lane >> fw.gl_shift, k = lane & (GL - 1);
    const int per_wave = RW * L * 2 + 16 + kWave * 8;  // the group's samples (+ one chunk of slack), 64 leaf sums
    uint16_t* smp = reinterpret_cast<uint16_t*>(s_dyn + (size_t)wv * per_wave);
    double* leaf_sum = reinterpret_cast<double*>(s_dyn + (size_t)wv * per_wave + (size_t)RW * L * 2 + 16);
    const int64_t n_groups = (rec.R + RW - 1) / RW;
    const int64_t wave0 = (int64_t)blockIdx.x * kFwWaves + wv;
    const int64_t nwaves = (int64_t)gridDim.x * kFwWaves;
    const fw_u4* __restrict__ p16 = reinterpret_cast<const fw_u4*>(fw.pool) + (fw.off0 >> 3);

    const bool leaf_live = k < plan.n_leaf;
    const int la = leaf_live ? plan.a[k] : 0, llen = leaf_live ? plan.len[k] : 0;

    fw_u4 pf[PFN];
    FwMeta nx;
    auto fetch = [&](int64_t grp) __attribute__((always_inline)) {
        const int64_t r0 = grp * RW;
        const int nrec = (int)(rec.R - r0 < RW ? rec.R - r0 : RW);
        const int chunks = nrec * CH;
        const fw_u4* src = p16 + r0 * CH;
#pragma unroll
        for (int t = 0; t < PFN; ++t) {
            const int c = t * kWave + lane;
            if ((t + 1) * kWave <= chunks) pf[t] = src[c];  // wave-uniform test
            else pf[t] = src[c < chunks ? c : 0];           // unconditional all the same: the loads issue back to back
        }
        const int64_t r = g < nrec ? r0 + g : r0;  // a lane past the end repeats the group's first record (never written)
        nx.bl = rec.baseline[r];
        nx.fb = (MODE == 0 && fw.fixed_bl) ? fw.fixed_bl[r] : 0.0;  // (MODE 2 is not launched with fixed baselines)
        nx.pol = rec.pol[r];
    };

    // A row is stored one group late, in front of the next prefetch: vmcnt counts loads and stores in order, so a store
    // issued after the prefetch would make the wait for the prefetched samples wait for the store's round trip too.
    // dwords computed here: 4 floats (basic) / 6 floats + q (width); the id columns follow in both layouts
    uint32_t prow_b[DO_B ? 4 : 1], prow_w[DO_W ? 8 : 1];
    FwCold cold;
    auto flush_row = [&](int64_t done) __attribute__((always_inline)) {  // the rows of group `done`, formed one turn ago
        const int64_t r = done * RW + g;
        if (k == 0 && r < rec.R) {
            if (DO_B) {
                uint32_t* at = reinterpret_cast<uint32_t*>(out + r * 36);
#pragma unroll
                for (int i = 0; i < 4; ++i) at[i] = prow_b[i];
                put_i64(at, 4, cold.ts);
                at[6] = cold.board | (cold.chan << 16);
                put_i64(at, 7, r);
            }
            if (DO_W) {
                uint32_t* at = reinterpret_cast<uint32_t*>((MODE == 2 ? out2 : out) + r * 52);
#pragma unroll
                for (int i = 0; i < 8; ++i) at[i] = prow_w[i];
                put_i64(at, 8, cold.ts);
                at[10] = cold.board | (cold.chan << 16);
                put_i64(at, 11, r);
            }
        }
    };
    auto fetch_cold = [&](int64_t r) __attribute__((always_inline)) {
        if (k == 0 && r < rec.R) {
            cold.ts = rec.ts[r];
            cold.board = (uint16_t)rec.board[r];
            cold.chan = (uint16_t)rec.chan[r];
        }
    };

#ifdef WFA_FW_TIMING
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int tit = 0;
#define FW_T(i) do { const long long t_now = (long long)__builtin_readcyclecounter(); tph[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define FW_T(i) do { } while (0)
#endif
    int64_t grp = uniform_i64(wave0);
    if (grp < n_groups) fetch(grp);
#ifdef WFA_FW_TIMING
    long long t_last = (long long)__builtin_readcyclecounter();
#endif
    for (; grp < n_groups; grp += nwaves) {
        const int64_t r0 = grp * RW;
        const int nrec = (int)(rec.R - r0 < RW ? rec.R - r0 : RW);
        const int chunks = nrec * CH;
#pragma unroll
        for (int t = 0; t < PFN; ++t) {
            const int c = t * kWave + lane;
            if ((t + 1) * kWave <= chunks || c < chunks) reinterpret_cast<fw_u4*>(smp)[c] = pf[t];
        }
        const FwMeta me = nx;
        FW_T(0);  // wait for the prefetch + LDS writes
        if (grp != wave0) flush_row(grp - nwaves);
        if (grp + nwaves < n_groups) fetch(grp + nwaves);  // in flight while this group is reduced
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        FW_T(1);  // row store + prefetch issue

        const bool valid = g < nrec;
        const int64_t r = r0 + g;
        const int pol = me.pol;
        double baseline = me.bl;
        if (MODE == 0 && fw.fixed_bl && me.fb == me.fb) baseline = me.fb;  // basic_features.py:143-146
        const bool known = pol == WFA_POL_NEGATIVE || pol == WFA_POL_POSITIVE;
        const bool wpos = pol == WFA_POL_POSITIVE_WAVE;
        const float b32 = (float)baseline;
        const TermKnown tk{b32, pol == WFA_POL_POSITIVE ? 0u : 0x80000000u};
        const TermWave tw{wpos ? 1.0 : -1.0, wpos ? -baseline : baseline};
        const bool all_known = __ballot(valid && !known) == 0, all_wave = __ballot(valid && known) == 0;
        const uint16_t* mine = smp + g * L;
        const fw_u4* q = reinterpret_cast<const fw_u4*>(mine + fw.c0 + la);  // c0 % 8 == 0, la % 8 == 0
        const int cmin = plan.min_len >> 3, cmax = (plan.max_len + 7) >> 3;
        double cs[kFwMaxChunks + 1];
        double res, res_b = 0.0;  // res: the reduction of MODE 0 / 1, in MODE 2 the clipped one (res_b the plain one)
        constexpr bool W = MODE != 0;
        if (MODE == 2) {
            if (all_known) res = leaf_sum_lane_both(q, llen, cmin, cmax, tk, cs, res_b);
            else if (all_wave) res = leaf_sum_lane_both(q, llen, cmin, cmax, tw, cs, res_b);
            else res = leaf_sum_lane_both(q, llen, cmin, cmax, TermMixed{tk, tw, known}, cs, res_b);
        } else if (all_known) res = leaf_sum_lane<W, W>(q, llen, cmin, cmax, tk, cs);
        else if (all_wave) res = leaf_sum_lane<W, W>(q, llen, cmin, cmax, tw, cs);
        else res = leaf_sum_lane<W, W>(q, llen, cmin, cmax, TermMixed{tk, tw, known}, cs);
        if (W && !leaf_live) {
#pragma unroll
            for (int m = 0; m <= kFwMaxChunks; ++m) cs[m] = 0.0;
        }
        FW_T(2);  // leaf sums
        // the leaves of a record combine along numpy's recursion tree
        double* ls = leaf_sum + g * GL;
        auto combine = [&](double leaf_value) __attribute__((always_inline)) {
            double root;
            if (plan.dpp_tree) {  // balanced: level l pairs the lanes 2^l apart (a + b == b + a bit for bit)
                root = leaf_value;
                if (plan.n_level > 0) root += fw_dpp_f64(root, 0);
                if (plan.n_level > 1) root += fw_dpp_f64(root, 1);
                if (plan.n_level > 2) root += fw_dpp_f64(root, 2);
                if (plan.n_level > 3) root += fw_dpp_f64(root, 3);
                if (GL != plan.n_leaf) root = __shfl(root, 0, GL);  // lanes without a leaf need it too
            } else {
                if (leaf_live) ls[k] = leaf_value;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (int lv = 0; lv < plan.n_level; ++lv) {
                    const int p = leaf_live ? plan.partner[lv][k] : -1;
                    double v = 0.0;
                    if (p >= 0) v = ls[k] + ls[p];
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (p >= 0) ls[k] = v;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                root = plan.n_leaf > 0 ? ls[0] : 0.0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            return 0.0 + (plan.n_leaf > 0 ? root : 0.0);
        };
        const double total = combine(res);                              // MODE 0: plain sum; MODE 1 / 2: clipped sum
        const double total_plain = MODE == 2 ? combine(res_b) : total;  // basic_features' area
        FW_T(3);  // tree

        // (the width part first: the per-chunk running sums die there, before the extremes of the basic part need registers)
        if (DO_W) {
            // quantile positions of x_i = max(signal_i, 0) (waveform_width_integral.py:180-231).  np.cumsum is sequential:
            // here each lane has the sequential sums of its own leaf and a scan over the record's lanes places them; a
            // decision closer than `tol` to a target is re-made by one lane in numpy's order.
            const double qsum = total;
            const bool ok = qsum > 0.0 && qsum <= 1.7976931348623157e308;  // finite and positive
            const double t_lo = fw.q_low * qsum, t_hi = fw.q_high * qsum;
            double incl = cs[kFwMaxChunks];
            double excl;
            if (GL <= 16) {  // a record's lanes sit in one row of 16: DPP shifts, no LDS round trips
                double o;
                if (GL > 1) { o = fw_row_shr_f64<1>(incl); if (k >= 1) incl += o; }
                if (GL > 2) { o = fw_row_shr_f64<2>(incl); if (k >= 2) incl += o; }
                if (GL > 4) { o = fw_row_shr_f64<4>(incl); if (k >= 4) incl += o; }
                if (GL > 8) { o = fw_row_shr_f64<8>(incl); if (k >= 8) incl += o; }
                excl = fw_row_shr_f64<1>(incl);
            } else {
                for (int d = 1; d < GL; d <<= 1) {
                    const double o = __shfl_up(incl, d, GL);
                    if (k >= d) incl += o;
                }
                excl = __shfl_up(incl, 1, GL);
            }
            if (k == 0) excl = 0.0;
            FW_T(5);  // scan
            const double eps = 8.0 * (double)L * 1.1102230246251565e-16;
            const double tt[2] = {t_lo, t_hi}, tol[2] = {eps * t_lo, eps * t_hi};
            int fnd[2];
            bool nr[2];
            if (all_known) leaf_crossings(q, la, llen, tk, cs, excl, tt, tol, fnd, nr);
            else if (all_wave) leaf_crossings(q, la, llen, tw, cs, excl, tt, tol, fnd, nr);
            else leaf_crossings(q, la, llen, TermMixed{tk, tw, known}, cs, excl, tt, tol, fnd, nr);
            FW_T(6);  // crossings
            if (!leaf_live) { fnd[0] = INT32_MAX; fnd[1] = INT32_MAX; nr[0] = false; nr[1] = false; }
            const int f_lo = fnd[0], f_hi = fnd[1];
            const int g_lo = group_min_i32(f_lo, GL), g_hi = group_min_i32(f_hi, GL);
            // lanes at or in front of the record's first crossing saw values around the target; the ones behind it did not
            bool amb = (nr[0] && (f_lo == g_lo || f_lo == INT32_MAX)) || (nr[1] && (f_hi == g_hi || f_hi == INT32_MAX));
            if (__ballot(g_lo == INT32_MAX || g_hi == INT32_MAX)) {  // no crossing at all: the last cumulative value decides
                const double last = __shfl(incl, GL - 1, GL);
                amb = amb || (g_lo == INT32_MAX && t_lo - last <= eps * t_lo) || (g_hi == INT32_MAX && t_hi - last <= eps * t_hi);
            }
            int amb_i = amb ? 1 : 0;
            amb_i = group_max_i32(amb_i, GL);
            int lo_i = g_lo == INT32_MAX ? L : g_lo;  // np.searchsorted returns len(cumsum)
            int hi_i = g_hi == INT32_MAX ? L : g_hi;
            FW_T(7);  // reductions
            if (amb_i && ok && valid && k == 0) {  // numpy's own order decides: k_width_ties re-walks the record
                const int slot = atomicAdd(fw.ties, 1);
                fw.ties[1 + slot] = (int32_t)r;
            }
            if (!ok) { lo_i = 0; hi_i = 0; }
            if (valid && k == 0) {
                uint32_t* row = prow_w;
                const double lo = (double)lo_i, hi = (double)hi_i;
                const double w = (double)(hi_i - lo_i > 0 ? hi_i - lo_i : 0);
                put_f32(row, 0, (float)(lo * fw.dt));
                put_f32(row, 1, (float)(hi * fw.dt));
                put_f32(row, 2, (float)(w * fw.dt));
                put_f32(row, 3, (float)lo);
                put_f32(row, 4, (float)hi);
                put_f32(row, 5, (float)w);
                put_f64(row, 6, qsum);
            }
        }
        if (DO_B) {
            // min / max of the raw samples over the height range: the record's lanes stride over it, four reads in flight
            int wmin = INT32_MAX, wmax = INT32_MIN;
            {
                const int span = fw.p1 - fw.p0;
                const int n_it = span > 0 ? (span + GL - 1) >> fw.gl_shift : 0;  // wave-uniform
                for (int it = 0; it < n_it; it += 4) {
                    int x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = fw.p0 + k + ((it + u) << fw.gl_shift);
                        x[u] = mine[i < fw.p1 ? i : fw.p0];  // a repeat of the first sample changes neither extreme
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        wmin = x[u] < wmin ? x[u] : wmin;
                        wmax = x[u] > wmax ? x[u] : wmax;
                    }
                }
            }
            // max |difference| over the record: each lane a run of chunks, two samples per operation.  Every lane reads
            // the same number of chunks (the last lane of a record repeats its last one, masked), so the reads batch.
            const int cpl = (CH + GL - 1) >> fw.gl_shift;
            const int cb = k * cpl, ce = cb + cpl < CH ? cb + cpl : CH;
            const fw_u4* rc = reinterpret_cast<const fw_u4*>(mine);
            fw_us2 dacc = {0, 0};
            uint32_t prev = cb > 0 && cb < CH ? reinterpret_cast<const uint32_t*>(mine)[cb * 4 - 1] : (uint32_t)mine[0] << 16;
            auto diff_chunk = [&](int t, const fw_u4& v) __attribute__((always_inline)) {
                fw_us2 cacc = {0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t di = v[i];  // (a bit_cast straight from the vector element reads element 0)
                    const uint32_t sh = __builtin_amdgcn_alignbit(di, prev, 16);  // the same samples, one place earlier
                    const fw_us2 cur = __builtin_bit_cast(fw_us2, di), old = __builtin_bit_cast(fw_us2, sh);
                    const fw_us2 df = __builtin_elementwise_max(cur, old) - __builtin_elementwise_min(cur, old);
                    cacc = __builtin_elementwise_max(cacc, df);
                    prev = di;
                }
                const uint32_t keep = cb + t < ce ? __builtin_bit_cast(uint32_t, cacc) : 0u;
                dacc = __builtin_elementwise_max(dacc, __builtin_bit_cast(fw_us2, keep));
            };
            auto chunk_at = [&](int t) { return rc[cb + t < CH ? cb + t : CH - 1]; };
            // four chunks at a time while the lanes have that many (wave-uniform test once per four: their LDS reads are
            // in flight together; a test per chunk left 13 dependent LDS round trips here), then the rest one by one
#pragma unroll
            for (int t0 = 0; t0 < kFwMaxChunks; t0 += 4) {
                if (t0 + 4 <= cpl) {
                    const fw_u4 v0 = chunk_at(t0), v1 = chunk_at(t0 + 1), v2 = chunk_at(t0 + 2), v3 = chunk_at(t0 + 3);
                    diff_chunk(t0, v0); diff_chunk(t0 + 1, v1); diff_chunk(t0 + 2, v2); diff_chunk(t0 + 3, v3);
                } else {
#pragma unroll
                    for (int t = t0; t < t0 + 4; ++t)
                        if (t < cpl) diff_chunk(t, chunk_at(t));
                }
            }
            int dmax = dacc.x > dacc.y ? dacc.x : dacc.y;
            wmin = group_min_i32(wmin, GL);
            wmax = group_max_i32(wmax, GL);
            dmax = group_max_i32(dmax, GL);
            if (valid && k == 0) {
                uint32_t* row = prow_b;
                float height = 0.f, amp = 0.f, area_f = 0.f, mad_f = 0.f;
                if (fw.p1 > fw.p0) {
                    double vmin, vmax;  // of `val` as the reference forms it (monotone in the sample)
                    if (known) {
                        const float lo = (float)wmin - b32, hi = (float)wmax - b32;
                        vmin = pol == WFA_POL_POSITIVE ? (double)lo : (double)-hi;
                        vmax = pol == WFA_POL_POSITIVE ? (double)hi : (double)-lo;
                    } else {
                        vmin = (double)wmin;
                        vmax = (double)wmax;
                    }
                    height = known ? (float)vmax : (wpos ? (float)(vmax - baseline) : (float)(baseline - vmin));
                    amp = (float)(vmax - vmin);
                }
                if (fw.c1 > fw.c0) area_f = (float)total_plain;
                if (L > 1) mad_f = (float)(double)dmax;
                put_f32(row, 0, height);
                put_f32(row, 1, amp);
                put_f32(row, 2, area_f);
                put_f32(row, 3, mad_f);
            }
        }
        fetch_cold(r);  // behind the prefetch in the queue, which the next turn waits for anyway
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        FW_T(4);  // the rest
#ifdef WFA_FW_TIMING
        ++tit;
#endif
    }
    if (grp != wave0) flush_row(grp - nwaves);
#ifdef WFA_FW_TIMING
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 1500))
        printf("fw mode %d block %d iters %d fallbacks %d cycles/iter: wait+ldsw %lld issue %lld leaf %lld tree %lld rest %lld (scan %lld cross %lld red %lld)\n",
               MODE, (int)blockIdx.x, tit & 0xffff, tit >> 16, tph[0] / (tit & 0xffff), tph[1] / (tit & 0xffff), tph[2] / (tit & 0xffff),
               tph[3] / (tit & 0xffff), tph[4] / (tit & 0xffff), tph[5] / (tit & 0xffff), tph[6] / (tit & 0xffff), tph[7] / (tit & 0xffff));
#endif
}

// The records k_features_leaf<1> could not call: one lane walks a record in numpy's order (np.cumsum is sequential) and
// rewrites the six position fields of its row.  A few per thousand records (cumulative values that tie with a target).
__global__ __launch_bounds__(64) void k_width_ties(FwParams fw, RecView rec, uint8_t* __restrict__ out) {
    const int n = fw.ties[0];
    const int CH = fw.L >> 3;
    const fw_u4* __restrict__ p16 = reinterpret_cast<const fw_u4*>(fw.pool) + (fw.off0 >> 3);
    for (int i = blockIdx.x * 64 + threadIdx.x; i < n; i += gridDim.x * 64) {
        const int64_t r = fw.ties[1 + i];
        uint32_t* row = reinterpret_cast<uint32_t*>(out + r * 52);
        const double qsum = __longlong_as_double((long long)((uint64_t)row[6] | ((uint64_t)row[7] << 32)));
        const double t_lo = fw.q_low * qsum, t_hi = fw.q_high * qsum;
        const int pol = rec.pol[r];
        const double baseline = rec.baseline[r];
        const bool known = pol == WFA_POL_NEGATIVE || pol == WFA_POL_POSITIVE;
        const bool wpos = pol == WFA_POL_POSITIVE_WAVE;
        const TermKnown tk{(float)baseline, pol == WFA_POL_POSITIVE ? 0u : 0x80000000u};
        const TermWave tw{wpos ? 1.0 : -1.0, wpos ? -baseline : baseline};
        const fw_u4* src = p16 + r * CH;
        int lo_i = -1, hi_i = -1;
        double c = 0.0;
        for (int ch = 0; ch < CH && hi_i < 0; ++ch) {
            const fw_u4 v = src[ch];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t x = (j & 1) ? v[j >> 1] >> 16 : v[j >> 1] & 0xffffu;
                const double sgl = known ? tk(x) : tw(x);
                c += sgl > 0.0 ? sgl : 0.0;
                if (lo_i < 0 && c >= t_lo) lo_i = ch * 8 + j;
                if (hi_i < 0 && c >= t_hi) hi_i = ch * 8 + j;
            }
        }
        if (lo_i < 0) lo_i = fw.L;
        if (hi_i < 0) hi_i = fw.L;
        const double lo = (double)lo_i, hi = (double)hi_i;
        const double w = (double)(hi_i - lo_i > 0 ? hi_i - lo_i : 0);
        put_f32(row, 0, (float)(lo * fw.dt));
        put_f32(row, 1, (float)(hi * fw.dt));
        put_f32(row, 2, (float)(w * fw.dt));
        put_f32(row, 3, (float)lo);
        put_f32(row, 4, (float)hi);
        put_f32(row, 5, (float)w);
    }
}

}  // namespace

// Uniform records (wfa_ctx::span_ok), uint16 pool, reduction length <= 8192: the lane-per-leaf kernels.  Returns
// false when the layout is outside that (the caller launches the general lane-per-record kernel).
static bool features_wave(wfa_ctx* c, int mode, const RecView& rec, FwParams fw, int n_sum, uint8_t* out, hipError_t* err,
                          uint8_t* out2 = nullptr) {
    *err = hipSuccess;
    if (!c->span_ok || c->span_L < 8 || c->span_L > kFwGroupSamples || c->opt.no_span) return false;
    if (fw.c0 % 8) return false;  // leaves start on 16-byte chunks of the staged record
    PwPlan plan;
    if (!pw_plan(plan, n_sum)) return false;
    if (c->pw_plan.ensure(sizeof(PwPlan)) != WFA_OK) return false;
    if (c->pw_plan_n != n_sum) {
        *err = hipMemcpyAsync(c->pw_plan.ptr, &plan, sizeof(plan), hipMemcpyHostToDevice, c->stream);
        if (*err == hipSuccess) *err = hipStreamSynchronize(c->stream);  // `plan` lives on this stack frame
        if (*err != hipSuccess) return true;
        c->pw_plan_n = n_sum;
    }
    fw.pool = c->pool_u16.as<uint16_t>();
    fw.off0 = c->span_off0;
    fw.L = c->span_L;
    if (mode >= 1) {
        if (rec.R >= INT32_MAX || c->fw_ties.ensure(((size_t)rec.R + 1) * sizeof(int32_t)) != WFA_OK) return false;
        fw.ties = c->fw_ties.as<int32_t>();
        *err = hipMemsetAsync(fw.ties, 0, sizeof(int32_t), c->stream);
        if (*err != hipSuccess) return true;
    }
    int sh = 0;
    while ((1 << sh) < plan.n_leaf) ++sh;                               // a lane per leaf
    while (sh < 6 && (int64_t)(kWave >> sh) * fw.L > kFwGroupSamples) ++sh;  // <= 8192 samples staged per wave
    fw.gl_shift = sh;
    const int RW = kWave >> sh;
    const size_t lds = (size_t)kFwWaves * ((size_t)RW * fw.L * 2 + 16 + kWave * 8);
    const int64_t n_groups = (rec.R + RW - 1) / RW;
    // persistent waves (each prefetches its next group): exactly the blocks that are resident together
    const bool small = (int64_t)RW * (fw.L >> 3) <= 13 * kWave;
    const void* fns[3][2] = {
        {reinterpret_cast<const void*>(k_features_leaf<0, 16>), reinterpret_cast<const void*>(k_features_leaf<0, 13>)},
        {reinterpret_cast<const void*>(k_features_leaf<1, 16>), reinterpret_cast<const void*>(k_features_leaf<1, 13>)},
        {reinterpret_cast<const void*>(k_features_leaf<2, 16>), reinterpret_cast<const void*>(k_features_leaf<2, 13>)}};
    const void* fn = fns[mode][small ? 1 : 0];
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kFwBlock, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    int64_t g = (n_groups + kFwWaves - 1) / kFwWaves;
    if (g > (int64_t)kFwCUs * per_cu) g = (int64_t)kFwCUs * per_cu;
    const PwPlan* dplan = c->pw_plan.as<PwPlan>();
    const dim3 grid((unsigned)g), block(kFwBlock);
#define WFA_FW_LAUNCH(M, P) hipLaunchKernelGGL((k_features_leaf<M, P>), grid, block, lds, c->stream, fw, rec, dplan, out, out2)
    if (mode == 0) { if (small) WFA_FW_LAUNCH(0, 13); else WFA_FW_LAUNCH(0, 16); }
    else if (mode == 1) { if (small) WFA_FW_LAUNCH(1, 13); else WFA_FW_LAUNCH(1, 16); }
    else { if (small) WFA_FW_LAUNCH(2, 13); else WFA_FW_LAUNCH(2, 16); }
#undef WFA_FW_LAUNCH
    if (mode >= 1 && hipGetLastError() == hipSuccess)
        hipLaunchKernelGGL(k_width_ties, dim3(256), dim3(64), 0, c->stream, fw, rec, mode == 2 ? out2 : out);
    *err = hipGetLastError();
    return true;
}

static void resolve_slice(int64_t start, int64_t end, int has_end, int L, int& lo, int& hi) {
    int64_t s = start;
    if (s < 0) { s += L; if (s < 0) s = 0; } else if (s > L) s = L;
    int64_t e = has_end ? end : (int64_t)L;
    if (e < 0) { e += L; if (e < 0) e = 0; } else if (e > L) e = L;
    lo = (int)s;
    hi = (int)(e < s ? s : e);
}

bool launch_basic_features_wave(wfa_ctx* c, const RecView& rec, const FeatParams& fp, uint8_t* out, hipError_t* err) {
    FwParams fw{};
    resolve_slice(fp.h0, fp.h1, fp.h_has_end, c->span_L, fw.p0, fw.p1);
    resolve_slice(fp.a0, fp.a1, fp.a_has_end, c->span_L, fw.c0, fw.c1);
    fw.fixed_bl = fp.fixed_bl;
    return features_wave(c, 0, rec, fw, fw.c1 - fw.c0, out, err);
}

// both tables from one read of the pool (MODE 2): only for the plugins' default reductions -- the area over the whole
// record and no per-channel fixed baselines -- where the two kernels walk the same leaves with the same baseline
bool launch_features_both_wave(wfa_ctx* c, const RecView& rec, const FeatParams& fp, const WidthParams& wp, uint8_t* out_basic,
                               uint8_t* out_width, hipError_t* err) {
    FwParams fw{};
    resolve_slice(fp.h0, fp.h1, fp.h_has_end, c->span_L, fw.p0, fw.p1);
    resolve_slice(fp.a0, fp.a1, fp.a_has_end, c->span_L, fw.c0, fw.c1);
    if (fp.fixed_bl || fw.c0 != 0 || fw.c1 != c->span_L) { *err = hipSuccess; return false; }
    fw.q_low = wp.q_low; fw.q_high = wp.q_high; fw.dt = wp.dt;
    return features_wave(c, 2, rec, fw, c->span_L, out_basic, err, out_width);
}

bool launch_width_integral_wave(wfa_ctx* c, const RecView& rec, const WidthParams& wp, uint8_t* out, hipError_t* err) {
    FwParams fw{};
    fw.q_low = wp.q_low; fw.q_high = wp.q_high; fw.dt = wp.dt;
    return features_wave(c, 1, rec, fw, c->span_L, out, err);
}

}  // namespace wfa
