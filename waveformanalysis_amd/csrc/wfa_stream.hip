// Streaming pass of the fused hit finder for uniform records (gfx950 / CDNA4):
//   baseline estimate + exact-integer Savitzky-Golay + threshold  ->  ordered run events per span
//
// Reference: WavePoolFilteredPlugin / _apply_filter_core (cpu/filtering.py:206-241) followed by
// ThresholdHitPlugin._build_hits_from_signal_matrix (cpu/hit_finder.py:329-366: mask = sig >= thr, runs of the mask),
// baseline = mean of the first samples (records_builder.py:243-257).  Arithmetic as in wfa_kernels.hip ("K7 fast
// path"): integer numerators Z = n . x, candidate <=> Z < zhi, the band of `margin` units below zhi is decided by the
// reference's float64 code.
//
// Shape (round 3).  Every integer / float64 / convert / DPP VALU instruction costs ~4.3 cycles per wave64 per SIMD
// (tools/op_rates.hip) and the round-2 form of this kernel issued ~435 of them per 2048-sample tile: 95 % of its run time
// was vector issue (profiles/r02_pmc_sq_counters.txt).  tools/mfma_stream_probe.hip showed what the tile body is worth
// without its bookkeeping: the same loads + the same FIR + the mask word written to LDS stream at 6.0 TB/s.  So the tile
// loop now does only what must be done per sample, and everything that is per record or per run happens once per span:
//   * a lane owns 32 consecutive samples of a tile (64 bytes, four 16-byte buffer loads, two tiles in flight);
//   * per tile: bias + halo exchange, the record's decision bound from a 256-byte LDS table, 32 numerators
//     (v_mad_i32_i16 + 5 v_dot2c), their signs gathered with one v_alignbit each -> one 32-bit mask word per lane,
//     stored to the span's bit image in LDS (record-major, odd word stride).  No validity masks, no neighbour bit, no
//     event bookkeeping in the loop; the band test exists only in the instance of the loop that spans with an
//     undecided integer take (3 % of them);
//   * per span, before the tiles: lane = record -- baseline (first 40 samples) and the exact float32 decision boundary;
//   * per span, after the tiles: lane = record evaluates the 2H edge samples with the polynomial-fit rows (samples read
//     again from L2), then lane = record segment (<= 32 mask words) fixes the words that hold edge samples / padding,
//     counts the transitions, one prefix sum places the span's events, and the lanes write them, ordered, into the
//     span's slot of the event buffer.  Start and end events alternate: event 2k / 2k + 1 are hit k of the span.
#include <cstdlib>
#include <type_traits>

#include "wfa_kernels.hpp"
#include "wfa_device.hpp"

namespace wfa {

namespace {

constexpr int kSpl = 32;                  // samples per lane per tile
constexpr int kTileSamples = kWave * kSpl;  // 2048
constexpr int kEvSlot = 2048;  // events a span may write (its slot of the event buffer)
constexpr int kMaxEventPos = 65504;  // largest record stride whose positions (0 .. S) fit the 16-bit field of an event
constexpr int kSegWords = 32;  // mask words one lane walks in the flush (a record longer than 1024 samples is split)

typedef unsigned short wfa_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int wfa_v4u __attribute__((ext_vector_type(4)));

struct StreamLds {  // per wave
    int32_t tot[kWave];  // baseline sums (BLW > 0)
    int32_t nz[kWave];   // -zhi: the addend that makes the sign of a numerator the candidate bit
    int32_t nb[kWave];   // undecided integers below zhi (0 almost always)
    uint32_t eb[kWave];  // edge bits of a record: bit e < H = sample e, bit H + e = sample L - H + e
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
//    own loads of the rare float64 paths) only make the wait stricter, never weaker.  Stores count too, in issue order:
//    the tile loop contains none (a store per tile put its write latency on the loop's critical path: 0.43 instead of
//    0.34 ms in the probe);
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
    asm volatile("s_waitcnt vmcnt(5) ; %0 %1 %2 %3 %4" : "+v"(d.q0), "+v"(d.q1), "+v"(d.q2), "+v"(d.q3), "+v"(d.peek) : : "memory");
}
__device__ __forceinline__ void tile_wait_all(Tile32& d) {
    asm volatile("s_waitcnt vmcnt(0) ; %0 %1 %2 %3 %4" : "+v"(d.q0), "+v"(d.q1), "+v"(d.q2), "+v"(d.q3), "+v"(d.peek) : : "memory");
}

__device__ __forceinline__ uint32_t udot2_acc(uint32_t pair, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(wfa_u2, pair), __builtin_bit_cast(wfa_u2, 0x00010001u), acc, false);
}
}  // namespace

// W: SG window (5..11).  BLW: 0 = records.baseline is given; 40 = baseline := mean of the first 40 samples, written
// back to records.baseline.  Records: uniform length L, stride S (multiple of 32, S - L < 32), contiguous from off0.
// A span = a.rs records (<= 64, and <= 64 flush segments); dynamic LDS: a.rs * a.wstride mask words per wave.
// 3 waves per SIMD: 168 vector registers; the kernel must not spill (the in-flight tile registers would be spilled with
// whatever they hold at that moment)
constexpr int kRunsOcc = 3;
#ifndef WFA_RUNS_BLOCK
#define WFA_RUNS_BLOCK 256
#endif
constexpr int kRunsBlock = WFA_RUNS_BLOCK, kRunsWaves = kRunsBlock / kWave;
template <int W, int BLW>
__global__ __launch_bounds__(kRunsBlock, kRunsOcc) void k_sg_runs32(RunsArgs a) {  // (threads per block, waves per SIMD)
    constexpr int H = W / 2;
    static_assert(W % 2 == 1 && W >= 5 && W <= 11, "halo of 6 samples per side");
    static_assert(BLW == 0 || BLW == 40, "in-stream baseline window");
    __shared__ __attribute__((aligned(16))) StreamLds s_lds[kRunsWaves];
    __shared__ int32_t etab[2 * H * W];
    extern __shared__ uint32_t s_words[];  // [kWavesPerBlock][rs * wstride]
    for (int k = threadIdx.x; k < 2 * H * W; k += kRunsBlock) etab[k] = a.itab[W + k];
    __syncthreads();
    const int lane = lane_id();
    StreamLds* __restrict__ lds = &s_lds[wave_in_block()];
    const int wstride = a.wstride;
    // per wave: the span's bit image, then the records' first 12 samples (6 dwords each, from the prologue) and -- when
    // a.dep -- the 64 bytes of the lane that holds each record's last samples (from the tile loop)
    const int per_wave = a.rs * wstride + a.rs * 6 + (a.dep ? a.rs * 16 : 0);
    uint32_t* __restrict__ words = s_words + wave_in_block() * per_wave;
    uint32_t* __restrict__ heads = words + a.rs * wstride;
    uint32_t* __restrict__ tails = heads + a.rs * 6;
    const bool dep = a.dep != 0;
    const int64_t wave0 = uniform_i64((int64_t)blockIdx.x * kRunsWaves + wave_in_block());
    const int64_t nwaves = (int64_t)gridDim.x * kRunsWaves;
    const int L = a.L;
    const int S = a.S;
    const int nw = S >> 5;   // mask words per record
    const int pad = S - L;  // < 32
    const bool positive = a.positive != 0;

    // first tap alone (a 16 x 16 + 32 multiply-add that takes the addend from another register: the 2-address
    // v_dot2c would need a copy of the addend per output), then the taps 1..W-1 as H pairs
    int c0 = a.itab[0];
    if (positive) c0 = -c0;  // Z = -(n.x - bias)
    uint32_t cq[H];
#pragma unroll
    for (int m = 0; m < H; ++m) {
        int n0 = a.itab[2 * m + 1], n1 = a.itab[2 * m + 2];
        if (positive) { n0 = -n0; n1 = -n1; }
        cq[m] = ((uint32_t)n0 & 0xffffu) | ((uint32_t)n1 << 16);
    }
    const uint32_t fillb = (positive ? 0u : 0xffffffffu) ^ 0x80008000u;
    const double den = (double)a.den;
    const double bias = 32768.0 * den;
    const double shift = positive ? bias : -bias;
    const double delta = a.delta;  // numerator units: |scipy's float64 chain - exact rational| * den, with head room
    const double rbl = 1.0 / (double)(BLW ? BLW : 1);
    // tile stepping of a lane's (record, position, mask word)
    const int step_q = kTileSamples / S, step_r = kTileSamples - step_q * S;
    const int step_w = step_q * wstride + (step_r >> 5);

    for (int64_t span = wave0; span < a.n_spans; span += nwaves) {
        const int64_t r0 = span * a.rs;
        const int nrec = (int)((a.R - r0) < a.rs ? (a.R - r0) : a.rs);
        const int64_t g_base = a.off0 + r0 * S;
        const int64_t r_lane = r0 + (lane < nrec ? lane : 0);
        const double thr_cur = a.thr[r_lane];
        const double bl_cur = BLW ? 0.0 : a.baseline[r_lane];
        bool any_band;
        {
            // lane = record of the span: its baseline (the first BLW samples: 80 bytes the tile loads read again a few
            // microseconds later, out of L2 / MALL) and its decision boundary, once per record
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
                    sum = udot2_acc(h.x, sum);
                    sum = udot2_acc(h.y, sum);
                    sum = udot2_acc(h.z, sum);
                    sum = udot2_acc(h.w, sum);
                    // the first 12 samples stay in LDS for the edge rows after the tiles
                    if (c == 0) { heads[lane * 6 + 0] = h.x; heads[lane * 6 + 1] = h.y; heads[lane * 6 + 2] = h.z; heads[lane * 6 + 3] = h.w; }
                    if (c == 1) { heads[lane * 6 + 4] = h.x; heads[lane * 6 + 5] = h.y; }
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
                const uint32_t* __restrict__ hp =
                    reinterpret_cast<const uint32_t*>(a.pool + g_base + (int64_t)(lane < nrec ? lane : 0) * S);
#pragma unroll
                for (int k = 0; k < 6; ++k) heads[lane * 6 + k] = hp[k];
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
            const int nb_l = zhi - 1 - (int)zl;  // integers strictly between zl and zh (0 almost always)
            lds->tot[lane] = tot_l;
            lds->nz[lane] = -zhi;
            lds->nb[lane] = nb_l;
            any_band = __ballot(nb_l > 0 && lane < nrec) != 0;
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

        // ---- the tiles: one mask word per lane and tile into the span's bit image ----
        // BAND: the span has a record with an undecided integer below its bound; only that instance of the loop tracks
        // the largest numerator of a half tile and knows the rare float64 path
        auto run_tiles = [&](auto band_tag) {
            constexpr bool BAND = decltype(band_tag)::value;
            int rl = (lane * kSpl) / S;
            int i0 = lane * kSpl - rl * S;  // multiple of 32; a lane's samples never straddle record slots
            int widx = rl * wstride + (i0 >> 5);
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
                // the lane that holds a record's last samples leaves its 64 bytes for the edge rows (2-3 lanes of a tile)
                if (dep && act && i0 == S - kSpl) {
                    wfa_v4u* __restrict__ tq = reinterpret_cast<wfa_v4u*>(tails + rl * 16);
                    tq[0] = tile.q0; tq[1] = tile.q1; tq[2] = tile.q2; tq[3] = tile.q3;
                }

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
                const int nzhi = lds->nz[rli];
                const int nband = BAND ? lds->nb[rli] : 0;

                // ---- numerators, two halves of 16 outputs ----
                uint32_t bits = 0;
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
                        if (BAND) umax = umax > (uint32_t)acc ? umax : (uint32_t)acc;
                    }
                    if (BAND) {
                        const uint32_t negband = 0u - (uint32_t)nband;  // Z - zhi in [-nband, -1]: undecided
                        if (__ballot(nband > 0 && umax >= negband) != 0) {  // (rare)
                            // valid outputs of the lane: not the H edge samples either side, not the padding (the flush
                            // overwrites those bits; the float64 code must not be asked for a sample behind the record)
                            const bool first = i0 == 0, last = i0 == S - kSpl;
                            const uint32_t vb_first = ~((1u << H) - 1u);
                            const uint32_t vb_last = (kSpl - pad - H) > 0 ? (0xffffffffu >> (32 - (kSpl - pad - H))) : 0u;
                            uint32_t vb = (first ? vb_first : 0xffffffffu) & (last ? vb_last : 0xffffffffu);
                            vb = act ? vb : 0u;
                            const uint32_t vbh = (vb >> (16 * h)) & 0xffffu;
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
                                const double baseline = BLW ? (double)lds->tot[rli] / (double)(BLW ? BLW : 1) : a.baseline[r0 + rli];
                                const double thr = a.thr[r0 + rli];
                                while (border) {
                                    const int jj = __ffs((int)border) - 1;
                                    border &= border - 1;
                                    const double w = src.at(i0 + 16 * h + jj);
                                    const double sig = positive ? (w - baseline) : (baseline - w);
                                    if (!(sig >= thr)) hb &= ~(1u << jj);
                                }
                            }
                        }
                    }
                    bits |= hb << (16 * h);
                }
                if (act) words[widx] = bits;

                // ---- next tile ----
                i0 += step_r;
                rl += step_q;
                widx += step_w;
                if (i0 >= S) { i0 -= S; ++rl; widx += wstride - nw; }
            };
            // The 6 samples in front of a tile (filter halo) are the previous tile's lane 63, read out of its registers
            // when it has been evaluated; the 16 bytes behind it come with the tile (`peek`).  (Scalar loads of those
            // bytes were tried: two scalar-cache misses to HBM per tile made the pass 3.5x slower.)
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
            for (int t = 0; t < T; t += 2) {
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
        };
        if (any_band) run_tiles(std::true_type{});
        else run_tiles(std::false_type{});
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // ---- after the tiles: one lane per record -- exact baseline, edge rows on the record's first / last W samples ----
        {
            uint32_t eb = 0;
            if (lane < nrec) {
                const int64_t r = r0 + lane;
                double baseline = bl_cur;
                if (BLW) {
                    baseline = (double)lds->tot[lane] / (double)(BLW ? BLW : 1);  // records_builder.py:243-257
                    a.baseline[r] = baseline;
                }
                int zhi_e, zlo_e;
                int_band(positive, baseline, thr_cur, (double)a.den_edge, 0.0, a.margin_edge, zhi_e, zlo_e);
                uint32_t border_e = 0;
                // 12 samples from the record's start, 12 from the even index at or in front of sample L - W
                const int te = (L - W) & ~1, o_tail = (L - W) - te;
                uint32_t hd[6], td[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) hd[k] = heads[lane * 6 + k];
                if (dep) {  // [te, te + 12) lies in the record's last lane: pad <= 20 (launch_sg_runs32)
                    const uint32_t* __restrict__ tl = tails + lane * 16 + ((te - (S - kSpl)) >> 1);
#pragma unroll
                    for (int k = 0; k < 6; ++k) td[k] = tl[k];
                } else {
                    const uint32_t* __restrict__ tp = reinterpret_cast<const uint32_t*>(a.pool + g_base + (int64_t)lane * S + te);
#pragma unroll
                    for (int k = 0; k < 6; ++k) td[k] = tp[k];
                }
#pragma unroll 1
                for (int side = 0; side < 2; ++side) {
                    int xw[W];
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const uint32_t d0 = side == 0 ? hd[k >> 1] : td[k >> 1], d1 = side == 0 ? hd[(k + 1) >> 1] : td[(k + 1) >> 1];
                        const uint32_t even = (k & 1) ? (d0 >> 16) : (d0 & 0xffffu);          // sample k of the row when o_tail == 0
                        const uint32_t odd = ((k + 1) & 1) ? (d1 >> 16) : (d1 & 0xffffu);     // sample k + 1
                        xw[k] = (int)((side == 1 && o_tail) ? odd : even);
                    }
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
                if (border_e) {  // rare: float64 reference code decides
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
            lds->eb[lane] = eb;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }

        // ---- flush: lane = segment of a record's mask words (a whole record when it has at most 32 words) ----
        // The tile loop left raw sign bits everywhere.  Here the words that hold the H edge samples either side get the
        // edge rows' bits, bits behind sample L - 1 are cleared, and the transitions of the result are the run events:
        // a prefix sum over the lanes (= over (record, position)) places them in the span's slot of the event buffer.
        {
            const int nseg = a.nseg, segw = a.segw;  // segments per record, words per segment (<= 32)
            const int rec_f = nseg == 1 ? lane : lane / nseg;
            const int sj = lane - rec_f * nseg;
            const bool live = rec_f < nrec;
            const int k0 = sj * segw;
            const int k1 = live ? ((k0 + segw) < nw ? (k0 + segw) : nw) : k0;
            uint32_t* __restrict__ wrow = words + (live ? rec_f : 0) * wstride;
            const uint32_t ebv = lds->eb[live ? rec_f : 0];
            const uint32_t eb_l = ebv & ((1u << H) - 1u), eb_r = ebv >> H;
            const int pR = L - H, kR = pR >> 5, shR = pR & 31;  // the right edge samples start at bit shR of word kR
            // word k of a record with its edge samples patched in and everything from sample L on cleared (idempotent)
            auto fixed = [&](int k, uint32_t w) {
                const int rem = pR - 32 * k;  // interior outputs of this word: bits below rem
                w = k == 0 ? (w & ~((1u << H) - 1u)) | eb_l : w;
                const uint32_t keep = rem >= 32 ? 0xffffffffu : (rem <= 0 ? 0u : ((1u << rem) - 1u));
                w &= keep;
                w |= k == kR ? (eb_r << shR) : 0u;
                w |= (k == kR + 1 && shR + H > 32) ? (eb_r >> ((32 - shR) & 31)) : 0u;
                return w;
            };
            // pass 1: fix, count transitions; wm = words that hold events, cm = their carry-in bits
            uint32_t wm = 0, cm = 0;
            int n_new = 0;
            uint32_t cin = (live && k0 > 0 && k0 < nw) ? (fixed(k0 - 1, wrow[k0 - 1]) >> 31) : 0u;
            auto walk = [&](auto uniform_tag) {
                constexpr bool UNI = decltype(uniform_tag)::value;  // one segment per record: every lane is at word kk
#pragma unroll 1
                for (int kk = 0; kk < segw; ++kk) {
                    const int k = UNI ? kk : k0 + kk;
                    if (k < k1) {
                        uint32_t w = wrow[k];
                        if (UNI) {
                            if (kk == 0 || kk >= kR) { w = fixed(kk, w); wrow[k] = w; }  // wave-uniform condition
                        } else {
                            w = fixed(k, w);
                            wrow[k] = w;
                        }
                        const uint32_t tr = w ^ ((w << 1) | cin);
                        cm |= cin << kk;
                        cin = w >> 31;
                        n_new += __popc(tr);
                        wm |= (uint32_t)(tr != 0u) << kk;
                    }
                }
            };
            if (nseg == 1) walk(std::true_type{});
            else walk(std::false_type{});
            // a run still open behind the record's last word closes at position S = L (pad == 0; with padding the cleared
            // bits behind L - 1 have closed it inside the word)
            const bool tail_ev = live && k1 == nw && k1 > k0 && cin != 0u;
            n_new += tail_ev ? 1 : 0;
            int total;
            const int dst0 = wave_excl_scan_i32(n_new, total);
            // every span owns kEvSlot events of the buffer: no allocation (a cursor word bumped by 19 531 waves sustains
            // ~90 returning atomics per microsecond: the waves queued for it for 10-25 us each)
            const unsigned long long base = (unsigned long long)span * kEvSlot;
            const bool fits = total <= kEvSlot && (int64_t)(base + kEvSlot) <= a.ev_cap;
            if (fits) {
                uint32_t* __restrict__ out = a.ev + base + dst0;
                const uint32_t tag = (uint32_t)rec_f << 16;
                uint32_t tr = 0, pos0 = 0;
                // pass 2: the words that hold events, in order; one event per lane and iteration
                while (__ballot((wm | tr) != 0u) != 0) {
                    if (tr == 0u && wm != 0u) {
                        const int kk = __ffs((int)wm) - 1;
                        wm &= wm - 1u;
                        const uint32_t w = wrow[k0 + kk];
                        tr = w ^ ((w << 1) | ((cm >> kk) & 1u));
                        pos0 = (uint32_t)(k0 + kk) * 32u;
                    }
                    if (tr != 0u) {
                        const int p = __ffs((int)tr) - 1;
                        tr &= tr - 1u;
                        *out++ = tag | (pos0 + (uint32_t)p);
                    }
                }
                if (tail_ev) *out = tag | (uint32_t)S;
            } else if (lane == 0) {
                atomicOr(a.flags, 1);  // more events than a span's slot holds: the caller takes the general route
            }
            if (lane == 0) {
                a.span_off[span] = (int64_t)base;
                a.span_cnt[span] = total >> 1;
                // hits of 64 consecutive spans: what k_runs_to_desc needs to place a span's rows without a scan launch
                if (fits) atomicAdd(&a.group_sum[span >> 6], (unsigned long long)(total >> 1));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

// events of a span -> hit descriptors (record, start, end, 0) in (record, start) order; one wave per span
__global__ __launch_bounds__(kBlock) void k_runs_to_desc(RunsParams rp, int64_t n_spans, int32_t rs,
                                                         int64_t* __restrict__ total_out, int64_t cap,
                                                         int4* __restrict__ desc) {
    const int64_t s = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    if (s >= n_spans) return;
    const int lane = lane_id();
    const int bad = *rp.flags;  // (tested after the loads below have been issued: one round trip instead of two)
    if (s == 0 && lane == 0 && rp.lit_cnt) *rp.lit_cnt = 0u;  // the list of hits the row kernel hands to the literal kernel
    const int cnt = rp.span_cnt[s];
    // first row of the span = hits of all spans in front of it: whole groups of 64 spans from the sums the streaming
    // kernel's flushes added up, the spans of its own group one per lane (this replaced a three-launch scan of span_cnt)
    const int64_t g = s >> 6;
    int64_t acc = 0;
    for (int64_t q = lane; q < g; q += kWave) acc += (int64_t)rp.group_sum[q];
    const int64_t sl = (g << 6) + lane;
    acc += sl < s ? (int64_t)rp.span_cnt[sl] : 0;
    const int64_t row0 = wave_sum_i64(acc);
    const uint2* __restrict__ ev = reinterpret_cast<const uint2*>(rp.ev + rp.span_off[s]);  // event counts are even
    if (bad != 0) {  // overflow: the events are incomplete, the caller redoes the pass; no rows until then
        if (s == n_spans - 1 && lane == 0) *total_out = 0;
        return;
    }
    if (s == n_spans - 1 && lane == 0) *total_out = row0 + cnt;
    for (int k = lane; k < cnt; k += kWave) {
        const uint2 e = ev[k];
        const int64_t row = row0 + k;
        if (cap == 0 || row < cap)
            desc[row] = make_int4((int)(s * rs + (e.x >> 16)), (int)(e.x & 0xffffu), (int)(e.y & 0xffffu), 0);
    }
}

int64_t sg_runs32_event_slot() { return kEvSlot; }

// records per span and LDS layout of a span's bit image for record stride S (multiple of 32):
// a record has S / 32 mask words and is walked by ceil(words / 32) lanes in the flush; a span holds as many records as
// 64 lanes can walk (at most 64); the word stride of a record in LDS is odd (lane = record reads hit 64 banks)
void sg_runs32_geometry(int32_t S, int32_t* rs, int32_t* wstride, int32_t* nseg, int32_t* segw) {
    const int nw = S / kSpl;
    const int ns = (nw + kSegWords - 1) / kSegWords;
    *nseg = ns;
    *segw = (nw + ns - 1) / ns;
    *rs = ns >= kWave ? 1 : kWave / ns;
    *wstride = nw | 1;
}

// LDS words of one wave's span: bit image + record heads (+ record tails)
int64_t sg_runs32_lds_words(int32_t rs, int32_t wstride, bool dep) { return (int64_t)rs * wstride + (int64_t)rs * 6 + (dep ? (int64_t)rs * 16 : 0); }

// May the tile loop deposit the records' last lanes in LDS (instead of the flush reading those samples again from
// memory)?  The 12 samples from the even index at or below L - W must lie in the record's last lane, and three blocks
// must still fit a CU's 160 KiB of LDS (a fourth wave per SIMD does not exist anyway: 168 registers).
bool sg_runs32_deposit(int32_t L, int32_t S, int32_t W, int32_t rs, int32_t wstride) {
    const int te = (L - W) & ~1;
    if (te < S - kSpl) return false;
    const int64_t block_bytes = (int64_t)kRunsWaves * (sg_runs32_lds_words(rs, wstride, true) * 4 + (int64_t)sizeof(StreamLds)) + 1024;
    return block_bytes * kRunsOcc * 4 / kRunsWaves <= 160 * 1024;
}

bool sg_runs32_supported(const SgParams& sg, int32_t L, int32_t S, int32_t bl_start, int32_t bl_end, bool fused_bl) {
    if (!sg.int_ok || sg.W < 5 || sg.W > 11 || !(sg.W & 1)) return false;
    if (S % kSpl != 0 || S - L < 0 || S - L >= kSpl || L < 64) return false;
    // run events are (record in span << 16) | position, and the tail event of a record sits at position L: every
    // position up to S must fit 16 bits.  Longer uniform records take the bitmap route (32-bit sample indices).
    if (S > kMaxEventPos) return false;  // (also: at most 64 flush segments of 32 words = 65 536 samples)
    if (S != L && (kSpl - (S - L)) < sg.W / 2) return false;  // the last lane of a record holds its H edge samples
    if (fused_bl && !(bl_start == 0 && bl_end == 40)) return false;
    return true;
}

hipError_t launch_sg_runs32(hipStream_t st, bool fused_baseline, const RunsArgs& a) {
    int64_t g = (a.n_spans + kRunsWaves - 1) / kRunsWaves;
    // one span per wave up to 64 rounds of the resident set: the dispatcher evens out the tail best with the smallest
    // blocks (v1725, 19 531 spans: 3 / 8 rounds / one span per wave = 0.514 / 0.502 / 0.499 ms in one run; persistent
    // waves with a hand-balanced last round of smaller spans: 0.53; spans of 32 records: 0.60-0.64 -- a span's fixed
    // costs, prologue + first tile + flush, are ~19 us of its 86)
    const int rounds = 64;
    const int64_t resident = (int64_t)rounds * 256 * kRunsOcc * 4 / kRunsWaves;
    if (g < 1) g = 1;
    if (g > resident) g = resident;
    const int grid = (int)g;
    const size_t dyn = (size_t)kRunsWaves * sg_runs32_lds_words(a.rs, a.wstride, a.dep != 0) * sizeof(uint32_t);
#define WFA_RUNS32(WW)                                                                                           \
    case WW:                                                                                                     \
        if (fused_baseline) hipLaunchKernelGGL((k_sg_runs32<WW, 40>), dim3(grid), dim3(kRunsBlock), dyn, st, a); \
        else hipLaunchKernelGGL((k_sg_runs32<WW, 0>), dim3(grid), dim3(kRunsBlock), dyn, st, a);                 \
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
                               int64_t* total_out, int64_t cap, int4* desc) {
    if (n_spans == 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_spans + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(k_runs_to_desc, dim3(grid), dim3(kBlock), 0, st, rp, n_spans, rs, total_out, cap, desc);
    return hipGetLastError();
}

}  // namespace wfa
