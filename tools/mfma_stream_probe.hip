// Probe for the next form of the streaming hit kernel: the Savitzky-Golay numerators of a 2048-sample tile on the
// matrix cores (v_mfma_i32_16x16x64_i8 on the two byte planes of the uint16 samples) instead of 6 VALU dot
// instructions per sample, against the same skeleton with the v_dot2 FIR of k_sg_runs32.  Both variants
//   * stream one private span per wave in 2048-sample tiles, two tiles in flight (hand-placed buffer loads),
//   * produce the candidate bit of every sample (sign of n . x + addend) with lane l owning samples [32 l, 32 l + 32)
//     of the tile -- the representation everything downstream of the FIR in k_sg_runs32 works on,
//   * either store that word (1 bit per sample: checked against the host) or only count its transitions.
// MFMA mapping (B = data, A = banded coefficient matrices, one per byte plane):
//   column c (0..15) of block b (0..7) of a tile = the 16 outputs at tile sample 128 c + 16 b + i; its K window is the
//   32 samples from 8 in front of them; lane (g, c) = 16 g + c supplies window samples 8 g .. 8 g + 7 = the aligned
//   16-byte chunk at tile byte 256 c + 32 b + 16 g - 16, loaded straight into the B operand (every chunk is loaded by
//   two lanes: L1 traffic, not HBM); A_lo[i][2 (8 + i - H + t)] = A_hi[i][.. + 1] = n[t].  D rows 4 g .. 4 g + 3 of
//   column c land in lane (g, c): outputs 128 c + 16 b + 4 g + r.  n . (x - 32896) = 256 D_hi + D_lo on bytes ^ 0x80.
//   The 8 nibbles of a lane go through LDS once per tile to become lane l's 32 consecutive bits.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_stream_probe.hip -o tools/mfma_stream_probe ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef short s2 __attribute__((ext_vector_type(2)));

constexpr int W = 11, H = 5;
constexpr int kTile = 2048;

struct Args {
    const uint16_t* pool;   // 16 readable bytes in front of it
    int64_t n_spans;
    int span_samples;       // multiple of 32
    int addend;             // candidate <=> n . (x - 32768) + addend < 0
    int n[W];
    uint32_t* bits;         // null: count only
    unsigned long long* count;
};

__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b), c, false);
}

// ---------------------------------------------------------------------------------------------------- MFMA variant
struct TileM {
    v4u q[8];
};
// MAP 0: block b of column c = outputs 128 c + 16 b (a lane's 8 chunks are 32 bytes apart: every cache line is touched by
// 4-5 of the 8 instructions); MAP 1: outputs 512 (b >> 1) + 32 c + 16 (b & 1) (an instruction reads 1 KiB contiguous, a
// line is touched by the two instructions of a block pair)
template <int MAP>
__device__ __forceinline__ void issue_m(TileM& d, v4u rsrc, uint32_t voff, uint32_t soff) {
    if (MAP == 1) {
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %8, %9, %10 offen\n\t"
            "buffer_load_dwordx4 %1, %8, %9, %10 offen offset:32\n\t"
            "buffer_load_dwordx4 %2, %8, %9, %10 offen offset:1024\n\t"
            "buffer_load_dwordx4 %3, %8, %9, %10 offen offset:1056\n\t"
            "buffer_load_dwordx4 %4, %8, %9, %10 offen offset:2048\n\t"
            "buffer_load_dwordx4 %5, %8, %9, %10 offen offset:2080\n\t"
            "buffer_load_dwordx4 %6, %8, %9, %10 offen offset:3072\n\t"
            "buffer_load_dwordx4 %7, %8, %9, %10 offen offset:3104"
            : "+v"(d.q[0]), "+v"(d.q[1]), "+v"(d.q[2]), "+v"(d.q[3]), "+v"(d.q[4]), "+v"(d.q[5]), "+v"(d.q[6]), "+v"(d.q[7])
            : "v"(voff), "s"(rsrc), "s"(soff)
            : "memory");
        return;
    }
    asm volatile(
        "s_nop 4\n\t"
        "buffer_load_dwordx4 %0, %8, %9, %10 offen\n\t"
        "buffer_load_dwordx4 %1, %8, %9, %10 offen offset:32\n\t"
        "buffer_load_dwordx4 %2, %8, %9, %10 offen offset:64\n\t"
        "buffer_load_dwordx4 %3, %8, %9, %10 offen offset:96\n\t"
        "buffer_load_dwordx4 %4, %8, %9, %10 offen offset:128\n\t"
        "buffer_load_dwordx4 %5, %8, %9, %10 offen offset:160\n\t"
        "buffer_load_dwordx4 %6, %8, %9, %10 offen offset:192\n\t"
        "buffer_load_dwordx4 %7, %8, %9, %10 offen offset:224"
        : "+v"(d.q[0]), "+v"(d.q[1]), "+v"(d.q[2]), "+v"(d.q[3]), "+v"(d.q[4]), "+v"(d.q[5]), "+v"(d.q[6]), "+v"(d.q[7])
        : "v"(voff), "s"(rsrc), "s"(soff)
        : "memory");
}
__device__ __forceinline__ void wait_but8(TileM& d) {
    asm volatile("s_waitcnt vmcnt(8) ; %0 %1 %2 %3 %4 %5 %6 %7"
                 : "+v"(d.q[0]), "+v"(d.q[1]), "+v"(d.q[2]), "+v"(d.q[3]), "+v"(d.q[4]), "+v"(d.q[5]), "+v"(d.q[6]), "+v"(d.q[7])
                 :
                 : "memory");
}
__device__ __forceinline__ void wait_all(TileM& d) {
    asm volatile("s_waitcnt vmcnt(0) ; %0 %1 %2 %3 %4 %5 %6 %7"
                 : "+v"(d.q[0]), "+v"(d.q[1]), "+v"(d.q[2]), "+v"(d.q[3]), "+v"(d.q[4]), "+v"(d.q[5]), "+v"(d.q[6]), "+v"(d.q[7])
                 :
                 : "memory");
}

template <int STORE, bool EARLY, int MAP>
__global__ __launch_bounds__(256, 3) void k_mfma(Args a) {
    __shared__ __attribute__((aligned(16))) uint32_t xs[4][64];
    __shared__ uint32_t sink[STORE == 2 ? 4 * 1664 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t span = (int64_t)blockIdx.x * 4 + wv;
    if (span >= a.n_spans) return;
    const int g = lane >> 4, c = lane & 15;
    // coefficient matrices: lane (g, m) holds A[m][16 g .. 16 g + 15]
    v4i alo, ahi;
    {
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        const int m = c;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = 16 * g + j, s = k >> 1, t = s - (8 + m - H);
            int coef = 0;
#pragma unroll
            for (int tt = 0; tt < W; ++tt) coef = (t == tt) ? a.n[tt] : coef;
            const uint32_t byte = (uint32_t)(coef & 0xff) << (8 * (j & 3));
            if (k & 1) hi[j >> 2] |= byte; else lo[j >> 2] |= byte;
        }
        alo = (v4i){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3]};
        ahi = (v4i){(int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
    int den = 0;
#pragma unroll
    for (int tt = 0; tt < W; ++tt) den += a.n[tt];
    const int nz = a.addend + 128 * den;  // n . (x - 32768) = (256 D_hi + D_lo) + 128 den
    const int span_bytes = a.span_samples * 2;
    const int T = (a.span_samples + kTile - 1) / kTile;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(a.pool + span * a.span_samples) - 16;
    v4u rsrc;
    rsrc.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)base);
    rsrc.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)((uint64_t)base >> 32) & 0xffffu));
    rsrc.z = (uint32_t)__builtin_amdgcn_readfirstlane(span_bytes + 16);
    rsrc.w = 0x00020000u;
    const uint32_t voff = (uint32_t)((MAP == 1 ? 64 : 256) * c + 16 * g);
    // after the exchange this lane owns tile samples [32 lane, 32 lane + 32): the nibbles of blocks 2 gd, 2 gd + 1 of the
    // four lanes of one column -- column lane >> 2, gd = lane & 3 (MAP 0) or column lane & 15, gd = lane >> 4 (MAP 1)
    const int gd = MAP == 1 ? (lane >> 4) : (lane & 3);
    uint32_t rot[4], msk[4];
#pragma unroll
    for (int gs = 0; gs < 4; ++gs) { rot[gs] = (uint32_t)(4 * (gd - gs)) & 31u; msk[gs] = 0x000F000Fu << (4 * gs); }
    uint32_t* xw = &xs[wv][4 * c + g];
    const v4u* xr = reinterpret_cast<const v4u*>(&xs[wv][4 * (MAP == 1 ? (lane & 15) : (lane >> 2))]);
    unsigned long long cnt = 0;
    uint32_t carry = 0;

    // the biased copies of a tile's chunks are the MFMA operands: the raw registers are free for the tile after next
    // as soon as the copies exist (EARLY: two whole tiles in flight while this one is evaluated)
    auto prep = [&](const TileM& tile, v4i (&bv)[8]) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            bv[b].x = (int)(tile.q[b].x ^ 0x80808080u); bv[b].y = (int)(tile.q[b].y ^ 0x80808080u);
            bv[b].z = (int)(tile.q[b].z ^ 0x80808080u); bv[b].w = (int)(tile.q[b].w ^ 0x80808080u);
        }
        // the copies must exist before the registers are handed to the next loads: volatile statements keep their order,
        // and a value that is an operand of one cannot be recomputed from the raw registers behind it
        asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
    };
    auto compute = [&](int t, const v4i (&bv)[8]) {
        uint32_t hb[2] = {0, 0};
#pragma unroll
        for (int half = 1; half >= 0; --half) {
            v4i dlo[4], dhi[4];
#pragma unroll
            for (int bb = 3; bb >= 0; --bb) {
                const int b = 4 * half + bb;
                dlo[bb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(alo, bv[b], (v4i){0, 0, 0, 0}, 0, 0, 0);
                dhi[bb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, bv[b], (v4i){0, 0, 0, 0}, 0, 0, 0);
            }
#pragma unroll
            for (int bb = 3; bb >= 0; --bb) {
                const int b = 4 * half + bb;
#pragma unroll
                for (int r = 3; r >= 0; --r) {
                    const int acc = (int)(((uint32_t)dhi[bb][r] << 8) + (uint32_t)dlo[bb][r]) + nz;
                    hb[b & 1] = __builtin_amdgcn_alignbit(hb[b & 1], (uint32_t)acc, 31);
                }
            }
        }
        // even blocks in the low half, odd blocks in the high half: nibble of block b at index (b >> 1) + 4 (b & 1)
        *xw = hb[0] | (hb[1] << 16);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const v4u x = *xr;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t xv[4] = {x.x, x.y, x.z, x.w};
        uint32_t bits = 0;
#pragma unroll
        for (int gs = 0; gs < 4; ++gs) bits |= __builtin_amdgcn_alignbit(xv[gs], xv[gs], rot[gs]) & msk[gs];
        if (STORE == 2) {
            sink[wv * 1664 + t * 64 + lane] = bits;  // the real kernel's shape: the word goes to LDS, the span's flush reads it
        } else if (STORE) {
            if (t < T) a.bits[(span * T + t) * 64 + lane] = bits;
        } else {
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)(carry << 31), (int)bits, 0x138, 0xf, 0xf, false);  // wave_shr:1
            cnt += __popc(bits ^ ((bits << 1) | (prev >> 31)));
            carry = (uint32_t)__builtin_amdgcn_readlane((int)bits, 63) >> 31;
        }
    };

    TileM ta{}, tb{};
    issue_m<MAP>(ta, rsrc, voff, 0u);
    if (EARLY) issue_m<MAP>(tb, rsrc, voff, (uint32_t)(kTile * 2));
    for (int t = 0; t < T; t += 2) {
        v4i bv[8];
        if (!EARLY) issue_m<MAP>(tb, rsrc, voff, (uint32_t)(t + 1) * (kTile * 2));
        wait_but8(ta);
        prep(ta, bv);
        if (EARLY) issue_m<MAP>(ta, rsrc, voff, (uint32_t)(t + 2) * (kTile * 2));
        compute(t, bv);
        if (!EARLY) issue_m<MAP>(ta, rsrc, voff, (uint32_t)(t + 2) * (kTile * 2));
        wait_but8(tb);
        prep(tb, bv);
        if (EARLY) issue_m<MAP>(tb, rsrc, voff, (uint32_t)(t + 3) * (kTile * 2));
        compute(t + 1, bv);  // a tile behind the span reads zeros
    }
    wait_all(ta);
    wait_all(tb);
    if (STORE == 2) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t w = 0;
        for (int k = 0; k < 25; ++k) w ^= sink[wv * 1664 + lane * 25 + k];  // lane = record: its 25 words
        if (w == 0x12345678u) atomicAdd(a.count, 1ull);
    }
    if (!STORE) {
        for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
        if (lane == 0) atomicAdd(a.count, cnt);
    }
}


// ---------------------------------------------------------------------------------------------------- load patterns only
// PAT 0: MAP 0 loads, 1: MAP 1 loads, 2: four fully coalesced 1-KiB loads per tile (no duplicates) + 4 dummy registers
template <int PAT>
__global__ __launch_bounds__(256, 3) void k_loads(Args a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t span = (int64_t)blockIdx.x * 4 + wv;
    if (span >= a.n_spans) return;
    const int g = lane >> 4, c = lane & 15;
    const int span_bytes = a.span_samples * 2;
    const int T = (a.span_samples + kTile - 1) / kTile;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(a.pool + span * a.span_samples) - 16;
    v4u rsrc;
    rsrc.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)base);
    rsrc.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)((uint64_t)base >> 32) & 0xffffu));
    rsrc.z = (uint32_t)__builtin_amdgcn_readfirstlane(span_bytes + 16);
    rsrc.w = 0x00020000u;
    const uint32_t voff = PAT == 2 ? (uint32_t)(16 * lane + 16) : (uint32_t)((PAT == 1 ? 64 : 256) * c + 16 * g);
    uint32_t acc = 0;
    auto issue = [&](TileM& d, uint32_t soff) {
        if (PAT == 2) {
            asm volatile(
                "s_nop 4\n\t"
                "buffer_load_dwordx4 %0, %4, %5, %6 offen\n\t"
                "buffer_load_dwordx4 %1, %4, %5, %6 offen offset:1024\n\t"
                "buffer_load_dwordx4 %2, %4, %5, %6 offen offset:2048\n\t"
                "buffer_load_dwordx4 %3, %4, %5, %6 offen offset:3072"
                : "+v"(d.q[0]), "+v"(d.q[1]), "+v"(d.q[2]), "+v"(d.q[3])
                : "v"(voff), "s"(rsrc), "s"(soff)
                : "memory");
        } else {
            issue_m<PAT>(d, rsrc, voff, soff);
        }
    };
    auto wait_one = [&](TileM& d) {
        if (PAT == 2) asm volatile("s_waitcnt vmcnt(4) ; %0 %1 %2 %3" : "+v"(d.q[0]), "+v"(d.q[1]), "+v"(d.q[2]), "+v"(d.q[3]) : : "memory");
        else wait_but8(d);
    };
    auto eat = [&](const TileM& d) {
#pragma unroll
        for (int b = 0; b < (PAT == 2 ? 4 : 8); ++b) acc ^= d.q[b].x ^ d.q[b].y ^ d.q[b].z ^ d.q[b].w;
    };
    TileM ta{}, tb{};
    issue(ta, 0u);
    for (int t = 0; t < T; t += 2) {
        issue(tb, (uint32_t)(t + 1) * (kTile * 2));
        wait_one(ta);
        eat(ta);
        issue(ta, (uint32_t)(t + 2) * (kTile * 2));
        wait_one(tb);
        eat(tb);
    }
    wait_all(ta);
    if (acc == 0x12345678u) atomicAdd(a.count, 1ull);
}

// ---------------------------------------------------------------------------------------------------- dot2 variant
struct TileD {
    v4u q0, q1, q2, q3, peek;
};
__device__ __forceinline__ void issue_d(TileD& d, v4u rsrc, uint32_t voff, uint32_t soff, uint32_t soff_next) {
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
__device__ __forceinline__ void wait_but5(TileD& d) {
    asm volatile("s_waitcnt vmcnt(5) ; %0 %1 %2 %3 %4" : "+v"(d.q0), "+v"(d.q1), "+v"(d.q2), "+v"(d.q3), "+v"(d.peek) : : "memory");
}
__device__ __forceinline__ void wait_all_d(TileD& d) {
    asm volatile("s_waitcnt vmcnt(0) ; %0 %1 %2 %3 %4" : "+v"(d.q0), "+v"(d.q1), "+v"(d.q2), "+v"(d.q3), "+v"(d.peek) : : "memory");
}
__device__ __forceinline__ uint32_t from_prev(uint32_t first, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ uint32_t from_next(uint32_t last, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)last, (int)v, 0x130, 0xf, 0xf, false);  // wave_shl:1
}

template <int STORE, bool EARLY, bool BAND>
__global__ __launch_bounds__(256, 3) void k_dot2(Args a) {
    __shared__ uint32_t sink[STORE == 2 ? 4 * 1664 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t span = (int64_t)blockIdx.x * 4 + wv;
    if (span >= a.n_spans) return;
    const int c0 = a.n[0];
    uint32_t cq[H];
#pragma unroll
    for (int m = 0; m < H; ++m) cq[m] = ((uint32_t)a.n[2 * m + 1] & 0xffffu) | ((uint32_t)a.n[2 * m + 2] << 16);
    const int nz = a.addend;
    const int span_bytes = a.span_samples * 2;
    const int T = (a.span_samples + kTile - 1) / kTile;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(a.pool + span * a.span_samples);
    v4u rsrc;
    rsrc.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)base);
    rsrc.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)((uint64_t)base >> 32) & 0xffffu));
    rsrc.z = (uint32_t)__builtin_amdgcn_readfirstlane(span_bytes);
    rsrc.w = 0x00020000u;
    const uint32_t voff = (uint32_t)lane * 64u;
    unsigned long long cnt = 0;
    uint32_t carry = 0;
    auto lane_dwords = [](const v4u& q, int ln) {
        v4u r;
        r.x = (uint32_t)__builtin_amdgcn_readlane((int)q.x, ln); r.y = (uint32_t)__builtin_amdgcn_readlane((int)q.y, ln);
        r.z = (uint32_t)__builtin_amdgcn_readlane((int)q.z, ln); r.w = (uint32_t)__builtin_amdgcn_readlane((int)q.w, ln);
        return r;
    };
    auto prep = [&](const TileD& tile, const v4u& hl, const v4u& hr, uint32_t (&E)[22]) {
        const uint32_t cur[16] = {tile.q0.x, tile.q0.y, tile.q0.z, tile.q0.w, tile.q1.x, tile.q1.y, tile.q1.z, tile.q1.w,
                                  tile.q2.x, tile.q2.y, tile.q2.z, tile.q2.w, tile.q3.x, tile.q3.y, tile.q3.z, tile.q3.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) E[3 + k] = cur[k] ^ 0x80008000u;
        E[0] = from_prev(hl.y ^ 0x80008000u, E[16]); E[1] = from_prev(hl.z ^ 0x80008000u, E[17]); E[2] = from_prev(hl.w ^ 0x80008000u, E[18]);
        E[19] = from_next(hr.x ^ 0x80008000u, E[3]); E[20] = from_next(hr.y ^ 0x80008000u, E[4]); E[21] = from_next(hr.z ^ 0x80008000u, E[5]);
        asm volatile("" : "+v"(E[0]), "+v"(E[1]), "+v"(E[2]), "+v"(E[3]), "+v"(E[4]), "+v"(E[5]), "+v"(E[6]), "+v"(E[7]), "+v"(E[8]), "+v"(E[9]),
                     "+v"(E[10]), "+v"(E[11]), "+v"(E[12]), "+v"(E[13]), "+v"(E[14]), "+v"(E[15]), "+v"(E[16]), "+v"(E[17]), "+v"(E[18]),
                     "+v"(E[19]), "+v"(E[20]), "+v"(E[21]));
    };
    auto compute = [&](int t, const uint32_t (&E)[22]) {
        uint32_t bits = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t Sh[13];
#pragma unroll
            for (int k = 0; k < 13; ++k) Sh[k] = __builtin_amdgcn_alignbit(E[8 * h + k + 1], E[8 * h + k], 16);
            uint32_t hb = 0, umax = 0;
#pragma unroll
            for (int jj = 15; jj >= 0; --jj) {
                const int ws = jj - H + 6;
                const uint32_t x0 = (ws & 1) == 0 ? E[8 * h + ws / 2] : Sh[(ws - 1) / 2];
                int acc;
                asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(acc) : "v"(x0), "s"(c0), "v"(nz));
#pragma unroll
                for (int m = 0; m < H; ++m) acc = sdot2((ws & 1) == 0 ? Sh[ws / 2 + m] : E[8 * h + (ws + 1) / 2 + m], cq[m], acc);
                hb = __builtin_amdgcn_alignbit(hb, (uint32_t)acc, 31);
                if (BAND) umax = umax > (uint32_t)acc ? umax : (uint32_t)acc;
            }
            if (BAND && __ballot(umax == 0x7fffffffu) != 0) hb ^= 1;  // keeps the band tracking of the real kernel alive
            bits |= (hb & 0xffffu) << (16 * h);
        }
        if (STORE == 2) {
            sink[wv * 1664 + t * 64 + lane] = bits;
        } else if (STORE) {
            if (t < T) a.bits[(span * T + t) * 64 + lane] = bits;
        } else {
            const uint32_t prev = from_prev(carry << 31, bits);
            cnt += __popc(bits ^ ((bits << 1) | (prev >> 31)));
            carry = (uint32_t)__builtin_amdgcn_readlane((int)bits, 63) >> 31;
        }
    };
    v4u carry_l = {0, 0, 0, 0};
    TileD ta{}, tb{};
    issue_d(ta, rsrc, voff, 0u, kTile * 2);
    if (EARLY) issue_d(tb, rsrc, voff, (uint32_t)(kTile * 2), (uint32_t)(2 * kTile * 2));
    for (int t = 0; t < T; t += 2) {
        uint32_t E[22];
        if (!EARLY) issue_d(tb, rsrc, voff, (uint32_t)(t + 1) * (kTile * 2), (uint32_t)(t + 2) * (kTile * 2));
        wait_but5(ta);
        prep(ta, carry_l, lane_dwords(ta.peek, 0), E);
        carry_l = lane_dwords(ta.q3, 63);
        if (EARLY) issue_d(ta, rsrc, voff, (uint32_t)(t + 2) * (kTile * 2), (uint32_t)(t + 3) * (kTile * 2));
        compute(t, E);
        if (!EARLY) issue_d(ta, rsrc, voff, (uint32_t)(t + 2) * (kTile * 2), (uint32_t)(t + 3) * (kTile * 2));
        wait_but5(tb);
        prep(tb, carry_l, lane_dwords(tb.peek, 0), E);
        carry_l = lane_dwords(tb.q3, 63);
        if (EARLY) issue_d(tb, rsrc, voff, (uint32_t)(t + 3) * (kTile * 2), (uint32_t)(t + 4) * (kTile * 2));
        compute(t + 1, E);
    }
    wait_all_d(tb);
    wait_all_d(ta);
    if (STORE == 2) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t w = 0;
        for (int k = 0; k < 25; ++k) w ^= sink[wv * 1664 + lane * 25 + k];
        if (w == 0x12345678u) atomicAdd(a.count, 1ull);
    }
    if (!STORE) {
        for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
        if (lane == 0) atomicAdd(a.count, cnt);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int span_samples = 64 * 800;
    const int64_t n_spans = argc > 1 ? atoll(argv[1]) : 19531;
    const int64_t N = n_spans * span_samples;
    const int n[W] = {-36, 9, 44, 69, 84, 89, 84, 69, 44, 9, -36};
    std::vector<uint16_t> h((size_t)N + 64);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (uint16_t)(8000 + (int)(s % 61) - 30 - ((s >> 20) % 97 == 0 ? 300 : 0));
    }
    uint16_t* d_pool; uint32_t* d_bits; unsigned long long* d_cnt;
    CK(hipMalloc(&d_pool, (size_t)(N + 64) * 2 + 256));
    CK(hipMemcpy(d_pool + 8, h.data(), (size_t)(N + 56) * 2, hipMemcpyHostToDevice));  // 16 bytes of front slack
    const int T = (span_samples + kTile - 1) / kTile;
    const size_t words = (size_t)n_spans * T * 64;
    CK(hipMalloc(&d_bits, words * 4));
    CK(hipMalloc(&d_cnt, 8));
    Args a{};
    a.pool = d_pool + 8; a.n_spans = n_spans; a.span_samples = span_samples;
    a.addend = -429 * (7985 - 32768);  // candidate <=> y < 7985
    for (int t = 0; t < W; ++t) a.n[t] = n[t];
    a.count = d_cnt;
    const unsigned grid = (unsigned)((n_spans + 3) / 4);
    // ---- correctness: every variant against the host on the first spans
    const int64_t check_spans = n_spans < 40 ? n_spans : 40;
    std::vector<uint32_t> got(words);
    struct Variant { const char* name; void (*store)(Args); void (*count)(Args); void (*lds)(Args); };
    const Variant variants[] = {
        {"mfma", k_mfma<1, false, 0>, k_mfma<0, false, 0>, k_mfma<2, false, 0>},
        {"mfma map1", k_mfma<1, false, 1>, k_mfma<0, false, 1>, k_mfma<2, false, 1>},
        {"mfma map1 early", k_mfma<1, true, 1>, k_mfma<0, true, 1>, k_mfma<2, true, 1>},
        {"dot2", k_dot2<1, false, true>, k_dot2<0, false, true>, k_dot2<2, false, true>},
        {"dot2 noband", k_dot2<1, false, false>, k_dot2<0, false, false>, k_dot2<2, false, false>},
        {"dot2 early noband", k_dot2<1, true, false>, k_dot2<0, true, false>, k_dot2<2, true, false>},
    };
    for (const Variant& v : variants) {
        a.bits = d_bits;
        CK(hipMemset(d_bits, 0, words * 4));
        hipLaunchKernelGGL(v.store, dim3(grid), dim3(256), 0, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), d_bits, words * 4, hipMemcpyDeviceToHost));
        long bad = 0, ones = 0;
        for (int64_t sp = 0; sp < check_spans; ++sp)
            for (int p = H; p < span_samples - H; ++p) {
                long acc = a.addend;
                for (int t = 0; t < W; ++t) acc += (long)n[t] * ((long)h[(size_t)sp * span_samples + p - H + t] - 32768);
                const int want = acc < 0;
                const int tile = p / kTile, q = p % kTile;
                const int have = (got[((size_t)sp * T + tile) * 64 + q / 32] >> (q % 32)) & 1;
                bad += want != have;
                ones += want;
            }
        printf("%-18s %ld mismatches on %lld interior samples (%ld candidates)\n", v.name, bad,
               (long long)(check_spans * (span_samples - 2 * H)), ones);
    }
    // ---- timing
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // dynamic LDS of 48 KiB per block = 3 blocks per CU = 3 waves per SIMD, the occupancy of the real kernel (its registers)
    for (const Variant& v : variants) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.store), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.count), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.lds), hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 1024));
    }
    {
        void (*pats[3])(Args) = {k_loads<0>, k_loads<1>, k_loads<2>};
        for (int rep = 0; rep < 2; ++rep)
            for (int pat = 0; pat < 3; ++pat)
                for (unsigned dyn = 0; dyn <= 48 * 1024; dyn += 48 * 1024) {
                    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pats[pat]), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
                    hipLaunchKernelGGL(pats[pat], dim3(grid), dim3(256), dyn, 0, a);
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(pats[pat], dim3(grid), dim3(256), dyn, 0, a);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("loads only, pattern %d %-6s %.4f ms  %.2f TB/s\n", pat, dyn ? "occ3" : "free", ms / 10, (double)N * 2 / (ms / 10 * 1e-3) / 1e12);
                }
    }
    for (int rep = 0; rep < 2; ++rep)
        for (const Variant& v : variants)
            for (int store = 0; store < 4; ++store) {
                if (store == 0) continue;  // (free occupancy of the count variant: not informative)
                a.bits = store == 1 ? d_bits : nullptr;
                // 3 blocks per CU: 48 KiB of dynamic LDS, or 24 KiB next to the 26 KiB sink of the lds variant
                const unsigned dyn = store == 1 ? 48 * 1024 : store == 2 ? 48 * 1024 : 24 * 1024;
                CK(hipMemset(d_cnt, 0, 8));
                auto launch = [&]() { hipLaunchKernelGGL(store == 1 ? v.store : store == 2 ? v.count : v.lds, dim3(grid), dim3(256), dyn, 0, a); };
                launch();
                CK(hipEventRecord(e0));
                for (int i = 0; i < 10; ++i) launch();
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("%-18s %-6s %-6s %.4f ms  %.2f TB/s\n", v.name, store == 1 ? "store" : store == 2 ? "count" : "lds", "occ3", ms / 10,
                       (double)N * 2 / (ms / 10 * 1e-3) / 1e12);
            }
    return 0;
}
