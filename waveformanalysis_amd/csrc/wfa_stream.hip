// Streaming pass of the fused hit finder for uniform records (gfx950 / CDNA4):
//   baseline estimate + exact-integer Savitzky-Golay + threshold  ->  ordered run events per span
//
// Reference: WavePoolFilteredPlugin / _apply_filter_core (cpu/filtering.py:206-241) followed by
// ThresholdHitPlugin._build_hits_from_signal_matrix (cpu/hit_finder.py:329-366: mask = sig >= thr, runs of the mask),
// baseline = mean of the first samples (records_builder.py:243-257).  Arithmetic as in wfa_kernels.hip ("K7 fast
// path"): integer numerators Z = n . x, candidate <=> Z < zhi, the band of `margin` units below zhi is decided by the
// reference's float64 code.
//
// Why this shape (measured on MI355X, tools/op_rates.hip): every integer / float64 / convert / DPP VALU instruction
// costs ~4.3 cycles per wave64 per SIMD, so the pass is bound by its instruction count per sample, not by HBM.
//   * a lane owns 32 consecutive samples (64 bytes, four 16-byte buffer loads): halo exchange, record tracking,
//     baseline, event emission are paid once per 32 samples;
//   * nothing is read twice: the baseline (first 40 samples) is summed from the tile registers of the lanes that
//     hold it (v_dot2_u32_u16 chain + one DPP shift), the threshold bound follows per lane through a 256-byte LDS
//     table; the wave-boundary halos are two scalar 16-byte loads per tile;
//   * no bitmap: the sign of (Z - zhi) is gathered with one v_alignbit per sample, transitions of the 32-bit lane
//     mask become (record, position) events, ordered by a DPP prefix sum, buffered in LDS and flushed once per span
//     (start and end events alternate, so event 2k / 2k+1 are hit k of the span);
//   * the 2H edge samples of a record use polynomial-fit rows, not the FIR.  The first / last lane of every record
//     deposit their samples in LDS; after the span's tiles one lane per record evaluates the edge rows exactly
//     (dense: 64 records at once).  Only if some edge sample is above threshold -- a pulse on the record boundary --
//     the wave streams the span a second time with those bits known.
#include <cstdlib>

#include "wfa_kernels.hpp"
#include "wfa_device.hpp"

namespace wfa {

namespace {

constexpr int kSpl = 32;                  // samples per lane per tile
constexpr int kTileSamples = kWave * kSpl;  // 2048
constexpr int kEvCap = 1024;   // events buffered per span and wave (typical span: ~220)
constexpr int kEvSlot = 2048;  // events a span may write: kEvCap + the 2 (H + 1) an edge patch can add per record
constexpr int kMaxEventPos = 65504;  // largest record stride whose positions (0 .. S) fit the 16-bit field of an event

typedef unsigned short wfa_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int wfa_v4u __attribute__((ext_vector_type(4)));
// constant address space: a uniform load through such a pointer is a scalar load (s_load_dwordx4) whatever the kernel
// stores elsewhere; the pool is not written while the pass runs
typedef __attribute__((address_space(4))) const wfa_v4u wfa_c_v4u;

struct StreamLds {  // per wave
    uint32_t ev[kEvCap];
    double thr[kWave];   // NaN thresholds / baselines stored as +inf: no hits
    double bl[kWave];    // given baselines (BLW == 0)
    int32_t tot[kWave];  // baseline sums (BLW > 0)
    int32_t nz[kWave];   // -zhi: the addend that makes the sign of a numerator the candidate bit
    int32_t nb[kWave];   // undecided integers below zhi (0 almost always)
    uint32_t eb[kWave];  // edge bits of the second pass
    uint32_t head[kWave][8];  // biased dwords: the first 16 samples of every record
    uint32_t tail[kWave][8];  // 16 samples that contain the last W samples of every record
};

// One tile = 64 bytes per lane in four 16-byte buffer loads, plus a fifth load of the 16 bytes behind the tile (same
// address in every lane: the right-hand halo, so that a tile depends on nothing that is loaded later).  Bytes behind the
// span's end read as 0.
//
// The loads are issued from inline asm and waited for with a hand-placed, counted s_waitcnt: hipcc's own waitcnt
// insertion cannot keep a prefetch in flight across the loop's back edge (it waits for vmcnt(0) right behind the issue),
// and with the loads at the top of an iteration the pass exposes the full HBM latency once per iteration (measured: 0.47
// of 0.99 ms).  Rules that make this safe (cdna_hip_programming.md section 5.7):
//  * the destination registers are read-write operands of the issue AND of the wait statement and are loop-carried
//    through an unroll-by-two, so they keep their physical registers; tools/audit_asm_loads.py checks in the generated
//    assembly that nothing touches them between an issue and its wait (part of `make`);
//  * vmcnt(5) = everything but the 5 youngest vector-memory operations has landed; more operations in between (hipcc's
//    own loads of the rare float64 paths) only make the wait stricter, never weaker;
//  * `s_nop 4` opens the issue: its scalar operands may have been written by the instruction before.
struct Tile32 {
    wfa_v4u q0, q1, q2, q3, peek;
};
__device__ __forceinline__ void tile_issue(Tile32& d, wfa_v4u rsrc, uint32_t voff, uint32_t soff, uint32_t soff_next) {
    asm volatile(
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, %5, %6, %7 offen\n\t"
        "buffer_load_dwordx4 %1, %5, %6, %7 offen offset:16\n\t"
        "buffer_load_dwordx4 %2, %5, %6, %7 offen offset:32\n\t"
        "buffer_load_dwordx4 %3, %5, %6, %7 offen offset:48\n\t"
        "buffer_load_dwordx4 %4, off, %6, %8"
        : "+v"(d.q0), "+v"(d.q1), "+v"(d.q2), "+v"(d.q3), "+v"(d.peek)
        : "v"(voff), "s"(rsrc), "s"(soff), "s"(soff_next)
        : "memory");
}
__device__ __forceinline__ void tile_wait_but5(Tile32& d) {
    asm volatile("s_waitcnt vmcnt(5)" : "+v"(d.q0), "+v"(d.q1), "+v"(d.q2), "+v"(d.q3), "+v"(d.peek) : : "memory");
}
__device__ __forceinline__ void tile_wait_all(Tile32& d) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(d.q0), "+v"(d.q1), "+v"(d.q2), "+v"(d.q3), "+v"(d.peek) : : "memory");
}

__device__ __forceinline__ uint32_t udot2_acc(uint32_t pair, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(wfa_u2, pair), __builtin_bit_cast(wfa_u2, 0x00010001u), acc, false);
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

}  // namespace

// W: SG window (5..11).  BLW: 0 = records.baseline is given; 40 = baseline := mean of the first 40 samples, written
// back to records.baseline.  Records: uniform length L, stride S (multiple of 32, S - L < 32), contiguous from off0.
// 3 waves per SIMD: 168 vector registers; the kernel must not spill (the in-flight tile registers would be spilled with
// whatever they hold at that moment)
constexpr int kRunsOcc = 3;
template <int W, int BLW>
__global__ __launch_bounds__(kBlock, kRunsOcc) void k_sg_runs32(RunsArgs a) {
    constexpr int H = W / 2;
    constexpr int NP = H + 1;
    static_assert(W % 2 == 1 && W >= 5 && W <= 11, "halo of 6 samples per side");
    static_assert(BLW == 0 || BLW == 40, "in-stream baseline window");
    __shared__ __attribute__((aligned(16))) StreamLds s_lds[kWavesPerBlock];
    __shared__ int32_t etab[2 * H * W];
    for (int k = threadIdx.x; k < 2 * H * W; k += kBlock) etab[k] = a.itab[W + k];
    __syncthreads();
    const int lane = lane_id();
#ifdef WFA_MEASURE
    const int dbg = a.dbg;  // measurement build only: 1 no events, 2 no deposits, 4 no filter arithmetic, 8 no bound update
#else
    constexpr int dbg = 0;
#endif
    StreamLds* __restrict__ lds = &s_lds[wave_in_block()];
    const int64_t wave0 = uniform_i64((int64_t)blockIdx.x * kWavesPerBlock + wave_in_block());
    const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
    const int L = a.L;
    const int S = a.S;
    const int pad = S - L;  // < 32
    const bool positive = a.positive != 0;

    uint32_t cpm[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        int n0 = a.itab[2 * m];
        int n1 = (2 * m + 1 < W) ? a.itab[2 * m + 1] : 0;
        if (positive) { n0 = -n0; n1 = -n1; }  // Z = -(n.x - bias)
        cpm[m] = ((uint32_t)n0 & 0xffffu) | ((uint32_t)n1 << 16);
    }
    // first tap alone (a 16 x 16 + 32 multiply-add that takes the addend from another register: the 2-address
    // v_dot2c would need a copy of the addend per output), then the taps 1..W-1 as H pairs
    int c0 = a.itab[0];
    if (positive) c0 = -c0;
    uint32_t cq[H];
#pragma unroll
    for (int m = 0; m < H; ++m) {
        int n0 = a.itab[2 * m + 1], n1 = a.itab[2 * m + 2];
        if (positive) { n0 = -n0; n1 = -n1; }
        cq[m] = ((uint32_t)n0 & 0xffffu) | ((uint32_t)n1 << 16);
    }
    const uint32_t fillb = (positive ? 0u : 0xffffffffu) ^ 0x80008000u;
    // threshold bound per lane (see "exact decision boundary" in do_tile)
    const double den = (double)a.den;
    const double bias = 32768.0 * den;
    const double shift = positive ? bias : -bias;
    const double delta = a.delta;  // numerator units: |scipy's float64 chain - exact rational| * den, with head room
    const double rbl = 1.0 / (double)(BLW ? BLW : 1);
    // tile stepping of a lane's (record, position)
    const int step_q = kTileSamples / S, step_r = kTileSamples - step_q * S;
    // tail deposit: 8 dwords E[d0e .. d0e + 7] hold the last W samples of a record (sample index o_tail + k of them)
    const int e0_tail = (kSpl - pad) - W + 6;  // E-sample index (E[0] = 6 samples before the lane's own) of sample L - W
    const int d0e = ((e0_tail >> 1) & ~1) < 14 ? ((e0_tail >> 1) & ~1) : 14;
    const int o_tail = e0_tail - 2 * d0e;
    const uint32_t vb_first = ~((1u << H) - 1u);
    const uint32_t vb_last = (kSpl - pad - H) > 0 ? (0xffffffffu >> (32 - (kSpl - pad - H))) : 0u;

    // per-record thresholds (and given baselines) of a span: one coalesced load, kept one span ahead
    auto load_thr = [&](int64_t span) {
        const int64_t r = span * a.rs + lane;
        return (span < a.n_spans && r < a.R) ? a.thr[r] : 0.0;
    };
    auto load_bl = [&](int64_t span) {
        const int64_t r = span * a.rs + lane;
        return (BLW == 0 && span < a.n_spans && r < a.R) ? a.baseline[r] : 0.0;
    };
    double thr_next = load_thr(wave0), bl_next = load_bl(wave0);

    for (int64_t span = wave0; span < a.n_spans; span += nwaves) {
        const int64_t r0 = span * a.rs;
        const int nrec = (int)((a.R - r0) < a.rs ? (a.R - r0) : a.rs);
        const int64_t g_base = a.off0 + r0 * S;
        const double thr_cur = thr_next, bl_cur = bl_next;
        thr_next = load_thr(span + nwaves);
        bl_next = load_bl(span + nwaves);
        {
            // lane = record of the span: its baseline (the first BLW samples: 80 bytes the tile loads read again a few
            // microseconds later, out of L2 / MALL) and its decision boundary, once per record -- inside the tile loop
            // the same float64 arithmetic ran once per tile for all lanes (113 of ~540 vector instructions per tile)
            const bool dead = !(thr_cur == thr_cur) || (BLW == 0 && !(bl_cur == bl_cur));
            const double thr_l = dead ? __builtin_huge_val() : thr_cur;
            int tot_l = 0;
            double b;
            if (BLW) {
                static_assert(BLW % 8 == 0 && BLW <= 64, "baseline window: whole 16-byte chunks");
                const uint4* __restrict__ hp =
                    reinterpret_cast<const uint4*>(a.pool + g_base + (int64_t)(lane < nrec ? lane : 0) * S);
                uint32_t sum = 0;
#pragma unroll
                for (int c = 0; c < BLW / 8; ++c) {
                    const uint4 h = hp[c];
                    sum = udot2_acc(h.x, 0x00010001u, sum);
                    sum = udot2_acc(h.y, 0x00010001u, sum);
                    sum = udot2_acc(h.z, 0x00010001u, sum);
                    sum = udot2_acc(h.w, 0x00010001u, sum);
                }
                tot_l = (int)sum;
                // tot / BLW, correctly rounded: reciprocal product + one FMA correction step (equal to the division for
                // every sum of 40 uint16 samples: tests/test_baseline_division_cpu.py)
                const double td = (double)tot_l;
                const double q0 = td * rbl;
                const double q1 = __builtin_fma(__builtin_fma(-(double)BLW, q0, td), rbl, q0);
                b = positive ? -q1 : q1;
            } else {
                b = positive ? -bl_cur : bl_cur;
            }
            // Exact decision boundary.  The reference masks  sig = +-(b - f32(y)) >= thr, i.e. (on the signed
            // quantities used here)  f32(y) <= v  with  v = +-b - thr.  Let lo <= v < hi be the adjacent float32
            // values around v: f32(y) <= v  <=>  y rounds to lo or below  <=>  y < (lo + hi) / 2.  With
            // y = Z / den + eps (|eps * den| <= delta) the mask is  Z < zt  for  zt = (lo + hi) / 2 * den  (exact
            // in float64), undecided only for integers within delta of zt: zl < Z < zh holds for at most one
            // integer, which the reference's float64 code decides (rare path in do_tile).
            const double v = b - thr_l;
            const float f = (float)v;
            const double fd = (double)f;
            const uint32_t u = __float_as_uint(f);
            const bool f_above = fd > v, f_pos = (u >> 31) == 0;
            const uint32_t u_lo = f_above ? (f_pos ? u - 1u : u + 1u) : u;   // one float32 towards -inf
            const uint32_t u_hi = f_above ? u : (f_pos ? u + 1u : u - 1u);   // one float32 towards +inf
            const double zt = ((double)__uint_as_float(u_lo) + (double)__uint_as_float(u_hi)) * (0.5 * den);
            double zl = floor(zt - delta), zh = ceil(zt + delta);
            // v = fl(+-b - thr) stands for the real number +-b - thr.  If it lies (almost) on a float32 value the
            // rounding of that subtraction decides the side: unless the subtraction was exact, take a band of one
            // float32 spacing either side and let the float64 code decide; the same for |v| < 1 (bit stepping
            // around zero)
            const double av = fabs(v);
            if (fabs(fd - v) <= av * 1e-11 || !(av >= 1.0)) {
                const bool exact = fabs(b) >= fabs(thr_l) && ((b - v) - thr_l) == 0.0 && av >= 1.0;  // Fast2Sum
                if (!exact) {
                    const double w = fmax(av, 1.0) * (den * 2.4e-7) + 1.0;  // > den * ulp_f32(v)
                    zl = floor(v * den - w);
                    zh = ceil(v * den + w);
                }
            }
            zh = fmin(fmax(zh + shift, -1073741824.0), 1073741824.0);  // NaN / -inf (no hits) -> -2^30
            zl = fmin(fmax(zl + shift, -1073741825.0), 1073741823.0);
            const int zhi = (int)zh;
            lds->thr[lane] = thr_l;
            lds->bl[lane] = bl_cur;
            lds->tot[lane] = tot_l;
            lds->nz[lane] = -zhi;
            lds->nb[lane] = zhi - 1 - (int)zl;  // integers strictly between zl and zh (0 almost always)
            lds->eb[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const int span_samples = nrec * S;  // multiple of 32
        const int span_bytes = span_samples * 2;
        const int T = (span_samples + kTileSamples - 1) / kTileSamples;
        const uint16_t* __restrict__ span_ptr = a.pool + g_base;
        // buffer descriptor of the span: base, stride 0, size in bytes (reads behind it return 0), raw dword format
        wfa_v4u rsrc;
        rsrc.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)span_ptr);
        rsrc.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)((uint64_t)span_ptr >> 32) & 0xffffu));
        rsrc.z = (uint32_t)__builtin_amdgcn_readfirstlane(span_bytes);
        rsrc.w = 0x00020000u;

        int n_ev = 0;
        {
            int rl = (lane * kSpl) / S;
            int i0 = lane * kSpl - rl * S;  // multiple of 32; a lane's samples never straddle record slots
            uint32_t carry31 = 0;
            // hl_r: the 16 bytes in front of the tile (x unused), hr_r: the 16 bytes behind it -- wave-uniform values
            auto do_tile = [&](int t, const Tile32& tile, const wfa_v4u& hl_r, const wfa_v4u& hr_r) {
                const uint32_t cur[16] = {tile.q0.x, tile.q0.y, tile.q0.z, tile.q0.w, tile.q1.x, tile.q1.y, tile.q1.z, tile.q1.w,
                                          tile.q2.x, tile.q2.y, tile.q2.z, tile.q2.w, tile.q3.x, tile.q3.y, tile.q3.z, tile.q3.w};
                // scalar: biased halo dwords (the last 6 samples before / the first 6 behind the tile)
                const bool t0 = t == 0;
                const uint32_t hl1 = t0 ? fillb : (hl_r.y ^ 0x80008000u), hl2 = t0 ? fillb : (hl_r.z ^ 0x80008000u),
                               hl3 = t0 ? fillb : (hl_r.w ^ 0x80008000u);
                const uint32_t hr0 = hr_r.x ^ 0x80008000u, hr1 = hr_r.y ^ 0x80008000u, hr2 = hr_r.z ^ 0x80008000u;
                const bool act = rl < nrec;
                const int rli = act ? rl : 0;
                const bool first = i0 == 0, last = i0 == S - kSpl;

                // ---- biased samples + halo (the raw tile registers die here) ----
                uint32_t E[22];
#pragma unroll
                for (int k = 0; k < 16; ++k) E[3 + k] = cur[k] ^ 0x80008000u;
                E[0] = dpp_from_prev_lane(hl1, E[16]);
                E[1] = dpp_from_prev_lane(hl2, E[17]);
                E[2] = dpp_from_prev_lane(hl3, E[18]);
                E[19] = dpp_from_next_lane(hr0, E[3]);
                E[20] = dpp_from_next_lane(hr1, E[4]);
                E[21] = dpp_from_next_lane(hr2, E[5]);

                // ---- threshold bound of the lane's record (from the span prologue) ----
                int nzhi = lds->nz[rli], nband = lds->nb[rli];

                // ---- deposits for the edge evaluation after the span ----
                if (!(dbg & 2)) {
                    if (first && act) {
                        *reinterpret_cast<uint4*>(&lds->head[rl][0]) = make_uint4(E[3], E[4], E[5], E[6]);
                        *reinterpret_cast<uint4*>(&lds->head[rl][4]) = make_uint4(E[7], E[8], E[9], E[10]);
                    }
                    if (last && act) {
                        uint2* dst = reinterpret_cast<uint2*>(&lds->tail[rl][0]);
#define WFA_TAIL_CASE(D)                                                                                     \
    case D:                                                                                                  \
        dst[0] = make_uint2(E[D], E[D + 1]); dst[1] = make_uint2(E[D + 2], E[D + 3]);                        \
        dst[2] = make_uint2(E[D + 4], E[D + 5]); dst[3] = make_uint2(E[D + 6], E[D + 7]);                    \
        break;
                        switch (d0e) {
                            WFA_TAIL_CASE(0) WFA_TAIL_CASE(2) WFA_TAIL_CASE(4) WFA_TAIL_CASE(6)
                            WFA_TAIL_CASE(8) WFA_TAIL_CASE(10) WFA_TAIL_CASE(12) WFA_TAIL_CASE(14)
                            default: break;
                        }
#undef WFA_TAIL_CASE
                    }
                }

                // ---- valid outputs of the lane: not the H edge samples either side, not the padding ----
                uint32_t vb = (first ? vb_first : 0xffffffffu) & (last ? vb_last : 0xffffffffu);
                vb = act ? vb : 0u;

                // ---- numerators, two halves of 16 outputs ----
                uint32_t bits = 0;
                if (dbg & 8) { nzhi = 0x40000000; nband = 0; }
                if (dbg & 4) bits = E[7] & E[12] & E[0] & E[21] & vb & 0x01000100u;
                else
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint32_t Sh[13];
#pragma unroll
                    for (int k = 0; k < 13; ++k) Sh[k] = __builtin_amdgcn_alignbit(E[8 * h + k + 1], E[8 * h + k], 16);
                    // Z - zhi of output 16h + jj (first tap carries -zhi: the sign is the candidate bit)
                    auto numer = [&](int jj, int addend) {
                        const int ws = jj - H + 6;  // first window sample, counted from E[8h]'s first sample (>= 1)
                        // (no inline-asm VOP3P first tap here: gfx950 needs wait states between a dot instruction and a
                        // different VALU instruction that reads its result, and hipcc pads only the instructions it knows)
                        // sample ws is the low half of E[8h + ws / 2] (ws even) or of Sh[(ws - 1) / 2] (ws odd); the pairs
                        // (ws + 1 + 2m, ws + 2 + 2m) then sit in the array of the other parity
                        const uint32_t x0 = (ws & 1) == 0 ? E[8 * h + ws / 2] : Sh[(ws - 1) / 2];
                        int acc;
                        asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(acc) : "v"(x0), "s"(c0), "v"(addend));
#pragma unroll
                        for (int m = 0; m < H; ++m) {
                            const uint32_t pair = (ws & 1) == 0 ? Sh[ws / 2 + m] : E[8 * h + (ws + 1) / 2 + m];
                            acc = sdot2_acc(pair, cq[m], acc);
                        }
                        return acc;
                    };
                    uint32_t hb = 0, umax = 0;
#pragma unroll
                    for (int jj = 15; jj >= 0; --jj) {
                        const int acc = numer(jj, nzhi);
                        hb = __builtin_amdgcn_alignbit(hb, (uint32_t)acc, 31);  // hb = (hb << 1) | sign(acc)
                        umax = umax > (uint32_t)acc ? umax : (uint32_t)acc;
                    }
                    const uint32_t vbh = (vb >> (16 * h)) & 0xffffu;
                    hb &= vbh;
                    const uint32_t negband = 0u - (uint32_t)nband;  // Z - zhi in [-nband, -1]: undecided
                    if (__ballot(nband > 0 && umax >= negband) != 0) {  // (rare)
                        // the numerators again (nothing of the common path is kept for this): an opaque copy of the
                        // addend stops the compiler from merging the two evaluations and hoisting the compares
                        int nz2 = nzhi;
                        asm volatile("" : "+v"(nz2));
                        uint32_t border = 0;
#pragma unroll
                        for (int jj = 0; jj < 16; ++jj) border |= (uint32_t)((uint32_t)numer(jj, nz2) >= negband) << jj;
                        border &= nband > 0 ? vbh : 0u;
                        if (border) {  // the reference's float64 arithmetic decides
                            WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(a.cold->pool, a.cold->sg, g_base + (int64_t)rli * S, L);
                            const double baseline = BLW ? (double)lds->tot[rli] / (double)(BLW ? BLW : 1) : lds->bl[rli];
                            const double thr = lds->thr[rli];
                            while (border) {
                                const int jj = __ffs((int)border) - 1;
                                border &= border - 1;
                                const double w = src.at(i0 + 16 * h + jj);
                                const double sig = positive ? (w - baseline) : (baseline - w);
                                if (!(sig >= thr)) hb &= ~(1u << jj);
                            }
                        }
                    }
                    bits |= hb << (16 * h);
                }
                // ---- run events: transitions of the mask inside the record ----
                {
                    const uint32_t prev = dpp_from_prev_lane(carry31 << 31, bits);
                    const uint32_t pb = first ? 0u : (prev >> 31);
                    uint32_t trans = bits ^ ((bits << 1) | pb);
                    const uint32_t tail_ev = (last && pad == 0) ? (bits >> 31) : 0u;  // run open at the record's end
                    const int cnt = __popc(trans) + (int)tail_ev;
                    carry31 = (uint32_t)__builtin_amdgcn_readlane((int)bits, 63) >> 31;
                    if (__ballot(cnt != 0) != 0 && !(dbg & 1)) {
                        int total;
                        int slot = n_ev + wave_excl_scan_i32(cnt, total);
                        const uint32_t evbase = ((uint32_t)rl << 16) | (uint32_t)i0;
                        if (n_ev + total <= kEvCap) {
                            while (trans) {
                                const int p = __ffs((int)trans) - 1;
                                trans &= trans - 1;
                                lds->ev[slot++] = evbase + (uint32_t)p;
                            }
                            if (tail_ev) lds->ev[slot] = evbase + (uint32_t)kSpl;
                        } else if (lane == 0) {
                            atomicOr(a.flags, 1);  // span does not fit the LDS buffer: the caller takes the general path
                        }
                        n_ev += total;
                    }
                }

                // ---- next tile ----
                i0 += step_r;
                rl += step_q;
                if (i0 >= S) { i0 -= S; ++rl; }
            };
            // The 6 samples in front of a tile (filter halo) are the previous tile's lane 63, read out of its registers
            // when it has been evaluated; the 16 bytes behind it (halo, and the baseline of a record that starts in the
            // last lane) come with the tile (`peek`).  (Scalar loads of those bytes were tried: two scalar-cache misses
            // to HBM per tile made the pass 3.5x slower.)
            const uint32_t voff = (uint32_t)lane * 64u;
            constexpr uint32_t kTileBytes = kTileSamples * 2;
            auto lane_dwords = [](const wfa_v4u& q, int ln) {
                wfa_v4u r;
                r.x = (uint32_t)__builtin_amdgcn_readlane((int)q.x, ln);
                r.y = (uint32_t)__builtin_amdgcn_readlane((int)q.y, ln);
                r.z = (uint32_t)__builtin_amdgcn_readlane((int)q.z, ln);
                r.w = (uint32_t)__builtin_amdgcn_readlane((int)q.w, ln);
                return r;
            };
            wfa_v4u carry_l = {0, 0, 0, 0};  // last 16 bytes of the previous tile (t == 0: unused)
            // ring of two tiles, unrolled by two: while one is evaluated the other (4 KiB per wave) is in flight
            Tile32 ta{}, tb{};
            tile_issue(ta, rsrc, voff, 0u, kTileBytes);
            for (int t = 0; t < ((dbg & 64) ? 0 : T); t += 2) {
                tile_issue(tb, rsrc, voff, (uint32_t)(t + 1) * kTileBytes, (uint32_t)(t + 2) * kTileBytes);
                tile_wait_but5(ta);
                do_tile(t, ta, carry_l, lane_dwords(ta.peek, 0));
                carry_l = lane_dwords(ta.q3, 63);
                tile_issue(ta, rsrc, voff, (uint32_t)(t + 2) * kTileBytes, (uint32_t)(t + 3) * kTileBytes);
                tile_wait_but5(tb);
                if (t + 1 < T) {
                    do_tile(t + 1, tb, carry_l, lane_dwords(tb.peek, 0));
                    carry_l = lane_dwords(tb.q3, 63);
                }
            }
            tile_wait_all(ta);  // the last issue (behind the span: nothing is fetched) must not outlive the registers
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // ---- after the tiles: one lane per record -- exact baseline, edge rows on the deposited samples ----
        uint32_t eb = 0;
        if (lane < nrec && !(dbg & 16)) {
            const int64_t r = r0 + lane;
            double baseline = bl_cur;
            if (BLW) {
                baseline = (double)lds->tot[lane] / (double)(BLW ? BLW : 1);  // records_builder.py:243-257
                a.baseline[r] = baseline;
            }
            int zhi_e, zlo_e;
            int_band(positive, baseline, thr_cur, (double)a.den_edge, 0.0, a.margin_edge, zhi_e, zlo_e);
            uint32_t border_e = 0;
            const uint16_t* hrow = reinterpret_cast<const uint16_t*>(&lds->head[lane][0]);
            const uint16_t* trow = reinterpret_cast<const uint16_t*>(&lds->tail[lane][0]) + o_tail;
#pragma unroll 1
            for (int side = (dbg & 128) ? 2 : 0; side < 2; ++side) {
                int xw[W];
                const uint16_t* row = side == 0 ? hrow : trow;
#pragma unroll
                for (int k = 0; k < W; ++k) xw[k] = (int)(row[k] ^ 0x8000u);
#pragma unroll 1
                for (int eh = 0; eh < H; ++eh) {
                    const int e = side * H + eh;
                    int acc = 0;
#pragma unroll
                    for (int k = 0; k < W; ++k) acc += etab[e * W + k] * xw[k];
                    const int ze = positive ? -acc : acc;
                    const bool m = ze < zhi_e;
                    eb |= (uint32_t)m << e;
                    border_e |= (uint32_t)(m && ze > zlo_e) << e;
                }
            }
            if (border_e && !(dbg & 256)) {  // rare: float64 reference code decides
                WaveSrc<WFA_SRC_SG_FUSED> src = make_src<WFA_SRC_SG_FUSED>(a.cold->pool, a.cold->sg, g_base + (int64_t)lane * S, L);
                while (border_e) {
                    const int e = __ffs((int)border_e) - 1;
                    border_e &= border_e - 1;
                    const double w = src.at(e < H ? e : L - 2 * H + e);
                    const double sig = positive ? (w - baseline) : (baseline - w);
                    if (!(sig >= thr_cur)) eb &= ~(1u << e);
                }
            }
        }
        if (dbg & 2) eb = 0;

        // ---- flush: the span's events with the edge samples patched in, one allocation ----
        // The tile loop treated the H edge samples either side of a record as "not a hit".  Around a record's start that
        // leaves at most one event to replace -- a START at sample H (interior bit H set) -- by the transitions of
        // [edge bits 0..H-1, bit H]; around its end an END at sample L - H by the transitions of [bit L-H-1, edge bits,
        // 0 behind the record].  One lane per record finds its slice of the (sorted) event list, counts its patched
        // events, a prefix sum places them, and the lane writes them out.  Pulse tails that reach the end of a record
        // are common: this has to cost nothing extra when there are none and little when there are.
        {
            const bool ok = n_ev <= kEvCap && !(dbg & 32);  // else: events were dropped, the caller takes the general route
            // slice [b, e) of record `lane` in lds->ev (sorted by record << 16 | position)
            int b = 0;
            if (ok) {
                const uint32_t key = (uint32_t)lane << 16;
                int lo = 0, hi = n_ev;  // first index with ev >= key
#pragma unroll 1
                for (int it = 0; it < 11; ++it) {
                    const int mid = (lo + hi) >> 1;
                    const bool go = lo < hi && lds->ev[mid < kEvCap ? mid : kEvCap - 1] < key;
                    lo = go ? mid + 1 : lo;
                    hi = (lo < hi && !go) ? mid : hi;
                }
                b = lo;
            }
            const int e = (int)dpp_from_next_lane((uint32_t)n_ev, (uint32_t)b);
            const int n_r = (lane < nrec && ok) ? e - b : 0;
            const uint32_t first_ev = n_r > 0 ? (lds->ev[b] & 0xffffu) : 0xffffu;
            const uint32_t last_ev = n_r > 0 ? (lds->ev[e - 1] & 0xffffu) : 0xffffu;
            const uint32_t bH = first_ev == (uint32_t)H ? 1u : 0u;        // interior bit at sample H
            const uint32_t bP = last_ev == (uint32_t)(L - H) ? 1u : 0u;   // interior bit at sample L - H - 1
            const uint32_t wL = (eb & ((1u << H) - 1u)) | (bH << H);     // samples 0 .. H
            const uint32_t tL = (wL ^ (wL << 1)) & ((2u << H) - 1u);     // transition in front of sample j
            const uint32_t wR = bP | ((eb >> H) << 1);                   // samples L-H-1 .. L-1 (then 0 at L)
            const uint32_t tR = ((wR ^ (wR << 1)) >> 1) & ((2u << H) - 1u);  // transition in front of sample L-H+j
            const int n_new = lane < nrec && ok ? n_r - (int)bH - (int)bP + __popc(tL) + __popc(tR) : 0;
            int total;
            const int dst0 = wave_excl_scan_i32(n_new, total);
            // every span owns kEvSlot events of the buffer: no allocation (a cursor word bumped by 19 531 waves sustains
            // ~90 returning atomics per microsecond: the waves queued for it for 10-25 us each)
            const unsigned long long base = (unsigned long long)span * kEvSlot;
            const bool fits = ok && total <= kEvSlot && (int64_t)(base + kEvSlot) <= a.ev_cap;
            if (fits) {
                uint32_t* __restrict__ out = a.ev + base + dst0;
                const uint32_t tag = (uint32_t)lane << 16;
                int k = 0;
                for (uint32_t t = tL; t; t &= t - 1) out[k++] = tag | (uint32_t)(__ffs((int)t) - 1);
                for (int i = b + (int)bH; i < e - (int)bP; ++i) out[k++] = lds->ev[i];
                for (uint32_t t = tR; t; t &= t - 1) out[k++] = tag | (uint32_t)(L - H + __ffs((int)t) - 1);
            } else if (lane == 0) {
                atomicOr(a.flags, ok ? 2 : 1);
            }
            if (lane == 0) {
                a.span_off[span] = (int64_t)base;
                a.span_cnt[span] = (ok ? total : n_ev) >> 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

// events of a span -> hit descriptors (record, start, end, 0) in (record, start) order; one wave per span
__global__ __launch_bounds__(kBlock) void k_runs_to_desc(RunsParams rp, int64_t n_spans, int32_t rs,
                                                         const int64_t* __restrict__ span_row0, int64_t cap,
                                                         int4* __restrict__ desc) {
    const int64_t s = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (s >= n_spans) return;
    if (*rp.flags != 0) return;  // overflow: the events are incomplete, the caller redoes the pass
    const int cnt = rp.span_cnt[s];
    const uint2* __restrict__ ev = reinterpret_cast<const uint2*>(rp.ev + rp.span_off[s]);  // event counts are even
    const int64_t row0 = span_row0[s];
    for (int k = lane_id(); k < cnt; k += kWave) {
        const uint2 e = ev[k];
        const int64_t row = row0 + k;
        if (cap == 0 || row < cap)
            desc[row] = make_int4((int)(s * rs + (e.x >> 16)), (int)(e.x & 0xffffu), (int)(e.y & 0xffffu), 0);
    }
}

int64_t sg_runs32_event_slot() { return kEvSlot; }

bool sg_runs32_supported(const SgParams& sg, int32_t L, int32_t S, int32_t bl_start, int32_t bl_end, bool fused_bl) {
    if (!sg.int_ok || sg.W < 5 || sg.W > 11 || !(sg.W & 1)) return false;
    if (S % kSpl != 0 || S - L < 0 || S - L >= kSpl || L < 64) return false;
    // run events are (record in span << 16) | position, and the tail event of a record sits at position L: every
    // position up to S must fit 16 bits.  Longer uniform records take the bitmap route (32-bit sample indices).
    if (S > kMaxEventPos) return false;
    if (S != L && (kSpl - (S - L)) < sg.W / 2) return false;  // the last lane of a record holds its H edge samples
    if (fused_bl && !(bl_start == 0 && bl_end == 40)) return false;
    return true;
}

hipError_t launch_sg_runs32(hipStream_t st, bool fused_baseline, const RunsArgs& a) {
    int64_t g = (a.n_spans + kWavesPerBlock - 1) / kWavesPerBlock;
    // one span per wave up to 64 rounds of the resident set: the dispatcher evens out the tail best with the smallest
    // blocks (v1725, 19 531 spans: 3 / 8 rounds / one span per wave = 0.514 / 0.502 / 0.499 ms in one run; persistent
    // waves with a hand-balanced last round of smaller spans: 0.53; spans of 32 records: 0.60-0.64 -- a span's fixed
    // costs, prologue + first tile + flush, are ~19 us of its 86)
    const int rounds = 64;
    const int64_t resident = (int64_t)rounds * 256 * kRunsOcc;
    if (g < 1) g = 1;
    if (g > resident) g = resident;
    const int grid = (int)g;
#define WFA_RUNS32(WW)                                                                                           \
    case WW:                                                                                                     \
        if (fused_baseline) hipLaunchKernelGGL((k_sg_runs32<WW, 40>), dim3(grid), dim3(kBlock), 0, st, a);       \
        else hipLaunchKernelGGL((k_sg_runs32<WW, 0>), dim3(grid), dim3(kBlock), 0, st, a);                       \
        break;
    switch (a.W) {
        WFA_RUNS32(5)
        WFA_RUNS32(7)
        WFA_RUNS32(9)
        WFA_RUNS32(11)
        default: return hipErrorInvalidValue;
    }
#undef WFA_RUNS32
    return hipGetLastError();
}

hipError_t launch_runs_to_desc(hipStream_t st, const RunsParams& rp, int64_t n_spans, int32_t rs,
                               const int64_t* span_row0, int64_t cap, int4* desc) {
    if (n_spans == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_spans + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(k_runs_to_desc, dim3(grid), dim3(kBlock), 0, st, rp, n_spans, rs, span_row0, cap, desc);
    return hipGetLastError();
}

}  // namespace wfa
