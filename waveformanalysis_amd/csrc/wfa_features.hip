// Per-record features on uniform records, one WAVE per record (BasicFeaturesPlugin records branch,
// cpu/basic_features.py:108-195; WaveformWidthIntegralPlugin, cpu/waveform_width_integral.py:83-231).
//
// The reference reduces with numpy, whose float64 results depend on the order of the additions; the rows must be bit
// identical.  The general kernels (wfa_kernels.hip: k_basic_features / k_width_integral) give every lane a whole record
// and walk it in numpy's order: correct for any layout, but a lane-per-record walk reads 16 bytes at a 1600-byte stride
// and runs 800 dependent float64 additions (0.12-0.15 of the HBM roofline).  Here the record is staged in LDS with
// coalesced 16-byte loads and numpy's order is mapped onto the wave:
//   np.sum  = pairwise_sum (umath/loops_utils.h.src): halves split at multiples of 8 down to leaves of <= 128 elements,
//             a leaf = 8 interleaved accumulators r_j = sum over m of x[a + 8 m + j], combined as
//             ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then its n % 8 tail added sequentially.
//             lane = (leaf, accumulator): 8 leaves x 8 accumulators per pass (an 800-sample record has exactly 8 leaves:
//             96 + 104 four times); every lane runs its own 12-13 additions in numpy's order, the 8 lanes of a leaf
//             combine with a DPP butterfly in numpy's parenthesisation, and the leaves combine along the recursion tree,
//             level by level, from a host-built plan (PwPlan).  Same additions, same operands, same order: bit exact.
//   min / max / max|diff| are order independent and exact on the raw integers.
//   np.cumsum + np.searchsorted: strictly sequential in numpy.  A wave scan adds in a different order, so its values may
//             differ from numpy's in the last bits; the quantile INDEX differs only if a cumulative value lies within
//             that rounding distance of the target.  The kernel takes the scan, checks the two values either side of
//             each crossing against the bound 8 L 2^-53 |target|, and re-walks the record sequentially in the (never
//             observed) case that one is closer.
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
    int32_t n_leaf, n_level, n, max_len, min_len, pad[3];
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
    return p.n_leaf <= 64 && p.n_level <= 8;
}

namespace {

constexpr int kFwBlock = 256;
constexpr int kFwMaxChunks = 16;   // 16-byte chunks per lane: staged group (<= 8192 samples per wave), leaf (<= 128 samples)
constexpr int kFwGroupSamples = 8192;

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
__device__ __forceinline__ void chunk_terms(const uint4& v, const F& f, double (&t)[8]) {
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t[2 * i] = f(d[i] & 0xffffu);
        t[2 * i + 1] = f(d[i] >> 16);
    }
    if (CLIP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_fmax(t[i], 0.0);  // a -0.0 term never changes a sum that starts at +0.0
    }
}

// One leaf of numpy's pairwise_sum per lane: q = the leaf's first chunk in LDS, len its length.  Returns the leaf's sum;
// with CUM also the running (sequential) sum after each chunk, cs[m], the partial tail counted as one more chunk.
template <bool CLIP, bool CUM, class F>
__device__ __forceinline__ double leaf_sum_lane(const uint4* q, int len, const F& f, double (&cs)[kFwMaxChunks + 1]) {
    const int cnt = len >> 3, nt = len & 7;
    double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double tot = 0.0;
#pragma unroll
    for (int m = 0; m < kFwMaxChunks; ++m) {
        if (m < cnt) {
            double t[8];
            chunk_terms<CLIP>(q[m], f, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = m == 0 ? t[j] : r[j] + t[j];  // r[j] = a[j]; r[j] += a[i + j]
            if (CUM) {
#pragma unroll
                for (int j = 0; j < 8; ++j) tot += t[j];
            }
        }
        if (CUM) cs[m] = tot;
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

// first sample of this lane's leaf at which the cumulative sum reaches `t` (INT32_MAX: not in this leaf).  excl = the
// cumulative value in front of the leaf.  near = the decision was closer than `tol` (the order of the additions could
// change it).
template <class F>
__device__ __forceinline__ int leaf_crossing(const uint4* q, int a, int len, const F& f, const double (&cs)[kFwMaxChunks + 1],
                                             double excl, double t, double tol, bool& near) {
    const int cnt = len >> 3, nt = len & 7, nch = cnt + (nt ? 1 : 0);
    int mm = 0;
    double start = 0.0;
#pragma unroll
    for (int m = 0; m < kFwMaxChunks; ++m) {  // cs is non-decreasing: the chunks in front of the crossing
        const bool below = m < cnt && excl + cs[m] < t;
        mm += below ? 1 : 0;
        start = below ? cs[m] : start;
    }
    if (nt && mm == cnt && excl + cs[kFwMaxChunks] < t) { mm = cnt + 1; start = cs[kFwMaxChunks]; }  // the partial chunk
    near = false;
    if (mm >= nch) {  // not in this leaf; its last value may still be too close to the target to call
        near = nch > 0 && t - (excl + start) <= tol;
        return INT32_MAX;
    }
    double tv[8];
    chunk_terms<true>(q[mm], f, tv);
    const int valid = mm < cnt ? 8 : nt;
    double run = excl + start;
    int found = INT32_MAX;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double before = run;
        run += tv[j];
        if (j < valid && found == INT32_MAX && run >= t) {
            found = a + mm * 8 + j;
            near = run - t <= tol || t - before <= tol;
        }
    }
    if (found == INT32_MAX) near = t - run <= tol;  // ended just below the target
    return found;
}

__device__ __forceinline__ int group_min_i32(int v, int gl) {
    for (int m = 1; m < gl; m <<= 1) { const int o = __shfl_xor(v, m, kWave); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int group_max_i32(int v, int gl) {
    for (int m = 1; m < gl; m <<= 1) { const int o = __shfl_xor(v, m, kWave); v = o > v ? o : v; }
    return v;
}

struct FwMeta {  // per-record columns, fetched one group ahead together with the samples
    double baseline;
    int64_t ts;
    int32_t pol, bc;
};

// MODE 0: BASIC_FEATURES_DTYPE rows (36 B); MODE 1: WAVEFORM_WIDTH_INTEGRAL_DTYPE rows (52 B).
// A wave takes a group of 64 >> gl_shift consecutive records; lane = (record of the group, leaf of the reduction).
template <int MODE>
__global__ __launch_bounds__(kFwBlock) void k_features_leaf(FwParams fw, RecView rec, const PwPlan* __restrict__ plan_g,
                                                            uint8_t* __restrict__ out) {
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
    const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + wv;
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const fw_u4* __restrict__ p16 = reinterpret_cast<const fw_u4*>(fw.pool) + (fw.off0 >> 3);

    const bool leaf_live = k < plan.n_leaf;
    const int la = leaf_live ? plan.a[k] : 0, llen = leaf_live ? plan.len[k] : 0;

    fw_u4 pf[kFwMaxChunks];
    FwMeta nx;
    auto fetch = [&](int64_t grp) __attribute__((always_inline)) {
        const int64_t r0 = grp * RW;
        const int nrec = (int)(rec.R - r0 < RW ? rec.R - r0 : RW);
        const int chunks = nrec * CH;
        const fw_u4* src = p16 + r0 * CH;
#pragma unroll
        for (int t = 0; t < kFwMaxChunks; ++t) {
            const int c = t * kWave + lane;
            pf[t] = src[c < chunks ? c : 0];  // unconditional: the loads of a group issue back to back
        }
        nx.baseline = 0.0; nx.ts = 0; nx.pol = 0; nx.bc = 0;
        if (g < nrec) {
            const int64_t r = r0 + g;
            nx.baseline = rec.baseline[r];
            if (MODE == 0 && fw.fixed_bl) {
                const double fb = fw.fixed_bl[r];
                if (fb == fb) nx.baseline = fb;  // basic_features.py:143-146
            }
            nx.pol = rec.pol[r];
            nx.ts = rec.ts[r];
            nx.bc = (int32_t)((uint32_t)(uint16_t)rec.board[r] | ((uint32_t)(uint16_t)rec.chan[r] << 16));
        }
    };

    int64_t grp = uniform_i64(wave0);
    if (grp < n_groups) fetch(grp);
    for (; grp < n_groups; grp += nwaves) {
        const int64_t r0 = grp * RW;
        const int nrec = (int)(rec.R - r0 < RW ? rec.R - r0 : RW);
        const int chunks = nrec * CH;
#pragma unroll
        for (int t = 0; t < kFwMaxChunks; ++t) {
            const int c = t * kWave + lane;
            if (c < chunks) reinterpret_cast<fw_u4*>(smp)[c] = pf[t];
        }
        const FwMeta me = nx;
        if (grp + nwaves < n_groups) fetch(grp + nwaves);  // in flight while this group is reduced
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        const bool valid = g < nrec;
        const int64_t r = r0 + g;
        const int pol = me.pol;
        const double baseline = me.baseline;
        const bool known = pol == WFA_POL_NEGATIVE || pol == WFA_POL_POSITIVE;
        const bool wpos = pol == WFA_POL_POSITIVE_WAVE;
        const float b32 = (float)baseline;
        const TermKnown tk{b32, pol == WFA_POL_POSITIVE ? 0u : 0x80000000u};
        const TermWave tw{wpos ? 1.0 : -1.0, wpos ? -baseline : baseline};
        const bool all_known = __ballot(valid && !known) == 0, all_wave = __ballot(valid && known) == 0;
        const uint16_t* mine = smp + g * L;
        const uint4* q = reinterpret_cast<const uint4*>(mine + fw.c0 + la);  // c0 % 8 == 0, la % 8 == 0
        double cs[kFwMaxChunks + 1];
        double res;
        constexpr bool W = MODE == 1;
        if (all_known) res = leaf_sum_lane<W, W>(q, llen, tk, cs);
        else if (all_wave) res = leaf_sum_lane<W, W>(q, llen, tw, cs);
        else res = leaf_sum_lane<W, W>(q, llen, TermMixed{tk, tw, known}, cs);
        // the leaves of a record combine along numpy's recursion tree
        double* ls = leaf_sum + g * GL;
        if (leaf_live) ls[k] = res;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int lv = 0; lv < plan.n_level; ++lv) {
            const int p = leaf_live ? plan.partner[lv][k] : -1;
            double v = 0.0;
            if (p >= 0) v = ls[k] + ls[p];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (p >= 0) ls[k] = v;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const double total = 0.0 + (plan.n_leaf > 0 ? ls[0] : 0.0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        if (MODE == 0) {
            // min / max of the raw samples over the height range: the record's lanes stride over it
            int wmin = INT32_MAX, wmax = INT32_MIN;
            for (int i = fw.p0 + k; i < fw.p1; i += GL) {
                const int x = mine[i];
                wmin = x < wmin ? x : wmin;
                wmax = x > wmax ? x : wmax;
            }
            // max |difference| over the record: each lane a run of chunks, two samples per operation
            const int cpl = (CH + GL - 1) >> fw.gl_shift;
            const int cb = k * cpl, ce = cb + cpl < CH ? cb + cpl : CH;
            const uint4* rc = reinterpret_cast<const uint4*>(mine);
            fw_us2 dacc = {0, 0};
            uint32_t prev = 0;
            if (cb < ce) prev = cb > 0 ? reinterpret_cast<const uint32_t*>(mine)[cb * 4 - 1] : (uint32_t)mine[0] << 16;
#pragma unroll
            for (int t = 0; t < kFwMaxChunks; ++t) {
                if (cb + t < ce) {
                    const uint4 v = rc[cb + t];
                    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t sh = __builtin_amdgcn_alignbit(d[i], prev, 16);  // the same samples, one place earlier
                        const fw_us2 cur = __builtin_bit_cast(fw_us2, d[i]), old = __builtin_bit_cast(fw_us2, sh);
                        const fw_us2 df = __builtin_elementwise_max(cur, old) - __builtin_elementwise_min(cur, old);
                        dacc = __builtin_elementwise_max(dacc, df);
                        prev = d[i];
                    }
                }
            }
            int dmax = dacc.x > dacc.y ? dacc.x : dacc.y;
            wmin = group_min_i32(wmin, GL);
            wmax = group_max_i32(wmax, GL);
            dmax = group_max_i32(dmax, GL);
            if (valid && k == 0) {
                uint32_t* row = reinterpret_cast<uint32_t*>(out + r * 36);
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
                if (fw.c1 > fw.c0) area_f = (float)total;
                if (L > 1) mad_f = (float)(double)dmax;
                put_f32(row, 0, height);
                put_f32(row, 1, amp);
                put_f32(row, 2, area_f);
                put_f32(row, 3, mad_f);
                put_i64(row, 4, me.ts);
                row[6] = (uint32_t)me.bc;
                put_i64(row, 7, r);
            }
        } else {
            // quantile positions of x_i = max(signal_i, 0) (waveform_width_integral.py:180-231).  np.cumsum is sequential:
            // here each lane has the sequential sums of its own leaf and a scan over the record's lanes places them; a
            // decision closer than `tol` to a target is re-made by one lane in numpy's order.
            const double qsum = total;
            const bool ok = qsum > 0.0 && qsum <= 1.7976931348623157e308;  // finite and positive
            const double t_lo = fw.q_low * qsum, t_hi = fw.q_high * qsum;
            double incl = cs[kFwMaxChunks];
            for (int d = 1; d < GL; d <<= 1) {
                const double o = __shfl_up(incl, d, GL);
                if (k >= d) incl += o;
            }
            double excl = __shfl_up(incl, 1, GL);
            if (k == 0) excl = 0.0;
            const double last = __shfl(incl, GL - 1, GL);
            const double eps = 8.0 * (double)L * 1.1102230246251565e-16;
            bool n_lo = false, n_hi = false;
            int f_lo, f_hi;
            if (all_known) {
                f_lo = leaf_crossing(q, la, llen, tk, cs, excl, t_lo, eps * t_lo, n_lo);
                f_hi = leaf_crossing(q, la, llen, tk, cs, excl, t_hi, eps * t_hi, n_hi);
            } else if (all_wave) {
                f_lo = leaf_crossing(q, la, llen, tw, cs, excl, t_lo, eps * t_lo, n_lo);
                f_hi = leaf_crossing(q, la, llen, tw, cs, excl, t_hi, eps * t_hi, n_hi);
            } else {
                const TermMixed tm{tk, tw, known};
                f_lo = leaf_crossing(q, la, llen, tm, cs, excl, t_lo, eps * t_lo, n_lo);
                f_hi = leaf_crossing(q, la, llen, tm, cs, excl, t_hi, eps * t_hi, n_hi);
            }
            if (!leaf_live) { f_lo = INT32_MAX; f_hi = INT32_MAX; n_lo = false; n_hi = false; }
            const int g_lo = group_min_i32(f_lo, GL), g_hi = group_min_i32(f_hi, GL);
            // lanes at or in front of the record's first crossing saw values around the target; the ones behind it did not
            bool amb = (n_lo && (f_lo == g_lo || f_lo == INT32_MAX)) || (n_hi && (f_hi == g_hi || f_hi == INT32_MAX));
            amb = amb || (g_lo == INT32_MAX && t_lo - last <= eps * t_lo) || (g_hi == INT32_MAX && t_hi - last <= eps * t_hi);
            int amb_i = amb ? 1 : 0;
            amb_i = group_max_i32(amb_i, GL);
            int lo_i = g_lo == INT32_MAX ? L : g_lo;  // np.searchsorted returns len(cumsum)
            int hi_i = g_hi == INT32_MAX ? L : g_hi;
            if (amb_i && ok && valid && k == 0) {  // numpy's own order decides (one lane, sequential)
                lo_i = -1; hi_i = -1;
                double c = 0.0;
                for (int i = 0; i < L && hi_i < 0; ++i) {
                    const uint32_t x = mine[i];
                    const double sgl = known ? tk(x) : tw(x);
                    c += sgl > 0.0 ? sgl : 0.0;
                    if (lo_i < 0 && c >= t_lo) lo_i = i;
                    if (hi_i < 0 && c >= t_hi) hi_i = i;
                }
                if (lo_i < 0) lo_i = L;
                if (hi_i < 0) hi_i = L;
            }
            if (!ok) { lo_i = 0; hi_i = 0; }
            if (valid && k == 0) {
                uint32_t* row = reinterpret_cast<uint32_t*>(out + r * 52);
                const double lo = (double)lo_i, hi = (double)hi_i;
                const double w = (double)(hi_i - lo_i > 0 ? hi_i - lo_i : 0);
                put_f32(row, 0, (float)(lo * fw.dt));
                put_f32(row, 1, (float)(hi * fw.dt));
                put_f32(row, 2, (float)(w * fw.dt));
                put_f32(row, 3, (float)lo);
                put_f32(row, 4, (float)hi);
                put_f32(row, 5, (float)w);
                put_f64(row, 6, qsum);
                put_i64(row, 8, me.ts);
                row[10] = (uint32_t)me.bc;
                put_i64(row, 11, r);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

}  // namespace

// Uniform records (wfa_ctx::span_ok), uint16 pool, reduction length <= 8192: the lane-per-leaf kernels.  Returns
// false when the layout is outside that (the caller launches the general lane-per-record kernel).
static bool features_wave(wfa_ctx* c, int mode, const RecView& rec, FwParams fw, int n_sum, uint8_t* out, hipError_t* err) {
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
    int sh = 0;
    while ((1 << sh) < plan.n_leaf) ++sh;                               // a lane per leaf
    while (sh < 6 && (int64_t)(kWave >> sh) * fw.L > kFwGroupSamples) ++sh;  // <= 8192 samples staged per wave
    fw.gl_shift = sh;
    const int RW = kWave >> sh;
    const size_t lds = (size_t)kWavesPerBlock * ((size_t)RW * fw.L * 2 + 16 + kWave * 8);
    const int64_t n_groups = (rec.R + RW - 1) / RW;
    int64_t g = (n_groups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > 256 * 4) g = 256 * 4;
    const PwPlan* dplan = c->pw_plan.as<PwPlan>();
    if (mode == 0) hipLaunchKernelGGL((k_features_leaf<0>), dim3((unsigned)g), dim3(kFwBlock), lds, c->stream, fw, rec, dplan, out);
    else hipLaunchKernelGGL((k_features_leaf<1>), dim3((unsigned)g), dim3(kFwBlock), lds, c->stream, fw, rec, dplan, out);
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

bool launch_width_integral_wave(wfa_ctx* c, const RecView& rec, const WidthParams& wp, uint8_t* out, hipError_t* err) {
    FwParams fw{};
    fw.q_low = wp.q_low; fw.q_high = wp.q_high; fw.dt = wp.dt;
    return features_wave(c, 1, rec, fw, c->span_L, out, err);
}

}  // namespace wfa
