// numpy's summation order, restated for device code (shared by the sample kernels and the hit-table kernels).
#pragma once

#include <hip/hip_runtime.h>

namespace wfa {

// numpy pairwise_sum (umath/loops_utils.h.src) of f(a) .. f(a+n-1), float64
template <typename F>
__device__ double np_pairwise_leaf(const F& f, int a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += f(a + i);
        return res;
    }
    double r0 = f(a), r1 = f(a + 1), r2 = f(a + 2), r3 = f(a + 3), r4 = f(a + 4), r5 = f(a + 5), r6 = f(a + 6),
           r7 = f(a + 7);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += f(a + i); r1 += f(a + i + 1); r2 += f(a + i + 2); r3 += f(a + i + 3);
        r4 += f(a + i + 4); r5 += f(a + i + 5); r6 += f(a + i + 6); r7 += f(a + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += f(a + i);
    return res;
}

template <typename F>
__device__ double np_pairwise_sum(const F& f, int a0, int n0) {
    if (n0 <= 128) return np_pairwise_leaf(f, a0, n0);
    constexpr int kDepth = 28;
    int sa[kDepth], sn[kDepth], sp[kDepth];
    double sl[kDepth];
    int top = 0;
    sa[0] = a0; sn[0] = n0; sp[0] = 0;
    double ret = 0.0;
    while (top >= 0) {
        const int a = sa[top], n = sn[top];
        if (n > 128) {  // descend left
            int n2 = n / 2;
            n2 -= n2 % 8;
            sp[top] = 1;
            ++top;
            sa[top] = a; sn[top] = n2; sp[top] = 0;
            continue;
        }
        ret = np_pairwise_leaf(f, a, n);
        --top;
        while (top >= 0) {
            if (sp[top] == 1) {  // left half done: keep it, descend right
                sl[top] = ret;
                sp[top] = 2;
                int n2 = sn[top] / 2;
                n2 -= n2 % 8;
                const int pa = sa[top], pn = sn[top];
                ++top;
                sa[top] = pa + n2; sn[top] = pn - n2; sp[top] = 0;
                break;
            }
            ret = sl[top] + ret;
            --top;
        }
    }
    return ret;
}

}  // namespace wfa
