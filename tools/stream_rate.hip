// Micro-benchmark: achievable HBM read rate of the access shapes the streaming kernels use, 2 GB buffer.
//   A  classic: consecutive workgroups read consecutive 16 B/lane chunks (grid-stride)
//   B  per-wave private streams: wave w reads its own contiguous span of SPAN bytes in TILE-byte tiles with
//      DEPTH tiles in flight (the k_sg_mask_span* shape), persistent grid of `waves` waves
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_rate.hip -o tools/stream_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_classic(const uint4* __restrict__ p, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}

// TILE16 = 16-byte units per lane per tile (1 -> 1 KB tiles, 2 -> 2 KB tiles)
template <int TILE16, int DEPTH>
__global__ __launch_bounds__(256) void k_spans(const uint4* __restrict__ p, size_t span16, size_t n_spans, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    unsigned acc = 0;
    const size_t tiles = span16 / (64 * TILE16);
    for (size_t s = wave; s < n_spans; s += nwaves) {
        const uint4* base = p + s * span16;
        uint4 ring[DEPTH][TILE16];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int k = 0; k < TILE16; ++k) ring[d][k] = base[(size_t)d * 64 * TILE16 + k * 64 + lane];
        for (size_t t = 0; t < tiles; t += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int k = 0; k < TILE16; ++k) { const uint4 v = ring[d][k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
                size_t nt = t + d + DEPTH;
                if (nt >= tiles) nt = tiles - 1;
#pragma unroll
                for (int k = 0; k < TILE16; ++k) ring[d][k] = base[nt * 64 * TILE16 + k * 64 + lane];
            }
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

// C: the k_sg_mask_span16 skeleton: 2 KB tiles (two 16-B loads per lane at +0/+16 of a 32-B lane slot), a ring of
// RING tiles, tile t consumed together with the first dwords of tile t+1 (readlane 0), VALU_OPS dependent integer
// ops per tile, one 2-byte store per lane per tile.
typedef short s2 __attribute__((ext_vector_type(2)));
template <int RING, int VALU_OPS, bool NEXT_DEP, bool STORE>
__global__ __launch_bounds__(256) void k_skel(const uint16_t* __restrict__ p, size_t span_samples, size_t n_spans,
                                             uint16_t* __restrict__ bm, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    unsigned acc = 0;
    const int T = (int)(span_samples / 1024);
    for (size_t s = wave; s < n_spans; s += nwaves) {
        const uint16_t* base = p + s * span_samples;
        uint16_t* bmb = bm + s * (span_samples / 16);
        uint4 ring[RING][2];
        auto load = [&](int t, uint4 (&r)[2]) {
            int tt = t < T ? t : T - 1;
            const uint16_t* q = base + (size_t)tt * 1024 + lane * 16;
            r[0] = *reinterpret_cast<const uint4*>(q);
            r[1] = *reinterpret_cast<const uint4*>(q + 8);
        };
#pragma unroll
        for (int d = 0; d < RING - 1; ++d) load(d, ring[d]);
        for (int t = 0; t < T; t += RING) {
#pragma unroll
            for (int d = 0; d < RING; ++d) {
                load(t + d + RING - 1, ring[(d + RING - 1) % RING]);
                const uint4 a = ring[d][0], b = ring[d][1];
                unsigned x = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
                if (NEXT_DEP) x ^= (unsigned)__builtin_amdgcn_readlane((int)ring[(d + 1) % RING][0].x, 0);
                int v = (int)x;
#pragma unroll
                for (int k = 0; k < VALU_OPS; ++k) v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, x), __builtin_bit_cast(s2, 0x00030005u + k), v, false);
                acc += (unsigned)v;
                if (STORE && t + d < T) bmb[(size_t)(t + d) * 64 + lane] = (uint16_t)v;
            }
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}


// D: lane-slot shapes: a lane owns SLOT bytes of every tile (SLOT/16 loads of 16 B at lane * SLOT + 16 k), two tiles
// loaded at the top of an iteration and consumed one after the other (the k_sg_runs32 shape at SLOT = 64).
// COAL: the same bytes read fully coalesced instead (load k of a tile covers 1 KiB: lane * 16 + 1024 k).
template <int SLOT, bool COAL, int VALU_OPS>
__global__ __launch_bounds__(256) void k_slot(const uint16_t* __restrict__ p, size_t span_bytes, size_t n_spans, unsigned* out) {
    constexpr int NL = SLOT / 16;
    constexpr int TILE = 64 * SLOT;
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    unsigned acc = 0;
    const int T = (int)(span_bytes / TILE);
    for (size_t s = wave; s < n_spans; s += nwaves) {
        const char* base = reinterpret_cast<const char*>(p) + s * span_bytes;
        for (int t = 0; t < T; t += 2) {
            uint4 a[NL], b[NL];
            const int t1 = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int k = 0; k < NL; ++k) a[k] = *reinterpret_cast<const uint4*>(base + (size_t)t * TILE + (COAL ? lane * 16 + 1024 * k : lane * SLOT + 16 * k));
#pragma unroll
            for (int k = 0; k < NL; ++k) b[k] = *reinterpret_cast<const uint4*>(base + (size_t)t1 * TILE + (COAL ? lane * 16 + 1024 * k : lane * SLOT + 16 * k));
            unsigned x = 0;
#pragma unroll
            for (int k = 0; k < NL; ++k) x ^= a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
            int v = (int)x;
#pragma unroll
            for (int k = 0; k < VALU_OPS * NL / 2; ++k) v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, x), __builtin_bit_cast(s2, 0x00030005u + k), v, false);
            acc += (unsigned)v;
            x = 0;
#pragma unroll
            for (int k = 0; k < NL; ++k) x ^= b[k].x ^ b[k].y ^ b[k].z ^ b[k].w;
            v = (int)x;
#pragma unroll
            for (int k = 0; k < VALU_OPS * NL / 2; ++k) v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, x), __builtin_bit_cast(s2, 0x00030005u + k), v, false);
            acc += (unsigned)v;
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <typename F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-46s %7.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 2000000000ull;
    const size_t n16 = bytes / 16;
    uint4* p; unsigned* out;
    hipMalloc(&p, bytes); hipMalloc(&out, 4);
    hipMemset(p, 1, bytes);
    timeit("A classic grid-stride, 256 CU x 8 WG", (double)bytes, [&] { hipLaunchKernelGGL(k_classic, dim3(2048), dim3(256), 0, 0, p, n16, out); });
    timeit("A classic grid-stride, 256 CU x 4 WG", (double)bytes, [&] { hipLaunchKernelGGL(k_classic, dim3(1024), dim3(256), 0, 0, p, n16, out); });
    const size_t span16 = 64 * 800 * 2 / 16;  // 64 records x 800 samples
    const size_t n_spans = n16 / span16;
    const double sb = (double)n_spans * span16 * 16;
    timeit("B spans 100 KB, 1 KB tiles, depth 4, 4096 waves", sb, [&] { hipLaunchKernelGGL((k_spans<1, 4>), dim3(1024), dim3(256), 0, 0, p, span16, n_spans, out); });
    timeit("B spans 100 KB, 2 KB tiles, depth 2, 4096 waves", sb, [&] { hipLaunchKernelGGL((k_spans<2, 2>), dim3(1024), dim3(256), 0, 0, p, span16, n_spans, out); });
    timeit("B spans 100 KB, 2 KB tiles, depth 4, 4096 waves", sb, [&] { hipLaunchKernelGGL((k_spans<2, 4>), dim3(1024), dim3(256), 0, 0, p, span16, n_spans, out); });
    timeit("B spans 100 KB, 2 KB tiles, depth 4, 8192 waves", sb, [&] { hipLaunchKernelGGL((k_spans<2, 4>), dim3(2048), dim3(256), 0, 0, p, span16, n_spans, out); });
    timeit("B spans 100 KB, 2 KB tiles, depth 8, 4096 waves", sb, [&] { hipLaunchKernelGGL((k_spans<2, 8>), dim3(1024), dim3(256), 0, 0, p, span16, n_spans, out); });
    timeit("B spans 100 KB, 2 KB tiles, depth 2, 2048 waves", sb, [&] { hipLaunchKernelGGL((k_spans<2, 2>), dim3(512), dim3(256), 0, 0, p, span16, n_spans, out); });
    {
        const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
        const size_t ss = 64 * 800, ns = (bytes / 2) / ss;
        uint16_t* bm; hipMalloc(&bm, bytes / 16);
        const double b2 = (double)ns * ss * 2;
#define SK(R, V, N, S, W) timeit("C skel ring " #R " valu " #V " nextdep " #N " store " #S " waves " #W, b2, [&] { \
            hipLaunchKernelGGL((k_skel<R, V, N, S>), dim3(W / 4), dim3(256), 0, 0, q, ss, ns, bm, out); });
        SK(3, 0, false, false, 4096) SK(3, 0, true, false, 4096) SK(3, 0, true, true, 4096)
        SK(3, 160, true, true, 4096) SK(4, 160, true, true, 4096) SK(3, 160, false, true, 4096)
        SK(2, 160, false, true, 4096) SK(3, 160, true, true, 8192) SK(4, 160, true, true, 8192)
        SK(3, 160, false, true, 8192) SK(2, 160, false, true, 8192)
    }
    {
        const uint16_t* q = reinterpret_cast<const uint16_t*>(p);
        const size_t sbytes = 64 * 800 * 2, ns = bytes / sbytes;   // 102400 B = 25 tiles of 4 KiB
        const double b2 = (double)ns * sbytes;
#define SL(SLOT, COAL, V, W) timeit("D slot " #SLOT " B/lane coalesced " #COAL " valu/2KB " #V " waves " #W, b2, [&] { \
            hipLaunchKernelGGL((k_slot<SLOT, COAL, V>), dim3(W / 4), dim3(256), 0, 0, q, sbytes, ns, out); });
        SL(64, false, 0, 3072) SL(64, false, 0, 4096) SL(64, true, 0, 3072) SL(64, true, 0, 4096)
        SL(32, false, 0, 3072) SL(32, false, 0, 4096) SL(32, true, 0, 4096) SL(128, false, 0, 3072) SL(128, true, 0, 3072)
        SL(64, false, 160, 3072) SL(64, true, 160, 3072) SL(32, false, 160, 4096)
    }
    // 16-record spans (25 KB): 4x as many, shorter streams
    const size_t span16b = 16 * 800 * 2 / 16;
    const size_t n_spans_b = n16 / span16b;
    timeit("B spans 25 KB, 2 KB tiles, depth 4, 4096 waves", (double)n_spans_b * span16b * 16, [&] { hipLaunchKernelGGL((k_spans<2, 4>), dim3(1024), dim3(256), 0, 0, p, span16b, n_spans_b, out); });
    return 0;
}
