// numpy's summation order, restated for device code (shared by the sample kernels and the hit-table kernels).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace wfa {

// numpy pairwise_sum (umath/loops_utils.h.src) of f(a) .. f(a+n-1); T = the array's dtype (double, or float for a
// float32 array: numpy accumulates float32 sums in float32)
template <typename T = double, typename F>
__device__ T np_pairwise_leaf(const F& f, int a, int n) {
    if (n < 8) {
        T res = (T)0;
        for (int i = 0; i < n; ++i) res += f(a + i);
        return res;
    }
    T r0 = f(a), r1 = f(a + 1), r2 = f(a + 2), r3 = f(a + 3), r4 = f(a + 4), r5 = f(a + 5), r6 = f(a + 6),
      r7 = f(a + 7);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += f(a + i); r1 += f(a + i + 1); r2 += f(a + i + 2); r3 += f(a + i + 3);
        r4 += f(a + i + 4); r5 += f(a + i + 5); r6 += f(a + i + 6); r7 += f(a + i + 7);
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += f(a + i);
    return res;
}

// numpy's recursion for n > 128: split at n2 = n/2 - (n/2) % 8, sum(left) + sum(right).  Iterative post-order walk:
// the only per-level state that must be kept is the left half's sum, in caller-provided scratch (`sl[level * stride]`,
// kPairwiseLevels levels: LDS, one column per thread -- register arrays indexed by the level cost 280 VGPRs);
// a frame's (offset, length) is recomputed from the root by following the path bits.
constexpr int kPairwiseLevels = 16;  // n <= 128 * 2^16 elements

__device__ __forceinline__ int np_pairwise_split(int n) {
    const int h = n / 2;
    return h - h % 8;
}

template <typename T = double, typename F>
__device__ T np_pairwise_sum(const F& f, int a0, int n0, double* sl, int stride) {
    if (n0 <= 128) return np_pairwise_leaf<T>(f, a0, n0);
    uint32_t path = 0;  // bit d set: the walk is in the right half at depth d
    int depth = 0, a = a0, n = n0;
    for (;;) {
        while (n > 128) {  // descend left
            path &= ~(1u << depth);
            ++depth;
            n = np_pairwise_split(n);
        }
        T ret = np_pairwise_leaf<T>(f, a, n);
        bool done = true;
        while (depth > 0) {
            --depth;
            int pa = a0, pn = n0;  // frame at `depth`
            for (int d = 0; d < depth; ++d) {
                const int n2 = np_pairwise_split(pn);
                if ((path >> d) & 1u) { pa += n2; pn -= n2; } else { pn = n2; }
            }
            if (!((path >> depth) & 1u)) {  // back from the left half: keep its sum, walk the right half
                sl[depth * stride] = (double)ret;
                path |= 1u << depth;
                const int n2 = np_pairwise_split(pn);
                a = pa + n2;
                n = pn - n2;
                ++depth;
                done = false;
                break;
            }
            ret = (T)sl[depth * stride] + ret;
        }
        if (done) return ret;
    }
}

}  // namespace wfa
