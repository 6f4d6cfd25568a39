// Hit-table stages on the device: hit merging (cpu/hit_merge.py) and event grouping
// (processing/event_grouping.py:286-471).  These are the reduce steps after the sample kernels: tables of
// 10^5..10^7 hit rows, sorted with rocPRIM's stable radix sort (through hipCUB) and reduced with scans and
// small per-group / per-cluster kernels.  Multi-key orders are built the way np.lexsort defines them:
// stable sorts from the least significant key to the most significant one.
#include <cstring>

#include <rocprim/block/block_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include "wfa_common.hpp"
#include "wfa_host.hpp"
#include "wfa_numpy.hpp"

namespace wfa {
namespace {

constexpr int kTB = 256;

__device__ __forceinline__ uint64_t ord_f64(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ uint64_t ord_i64(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kTB - 1) / kTB); }

// ---- scratch slots of wfa_ctx::ht ------------------------------------------------------------------------
enum Slot {
    S_TS, S_POS, S_START, S_END, S_DT, S_BOARD, S_CHAN, S_RID, S_HEIGHT, S_INTEGRAL,
    S_ABS0, S_ABS1, S_K0, S_K1, S_K2, S_K3, S_K4, S_KTMP0, S_KTMP1, S_PERM0, S_PERM1, S_CUB,
    S_F0, S_F1, S_FLAG, S_ID, S_OUT0, S_OUT1, S_OUT2, S_OUT3, S_OUT4, S_OUT5, S_OUT6, S_OUT7, S_CNT, S_CNT2, S_SG, S_CSV, S_N
};
static_assert(S_N <= 40, "wfa_ctx::ht is too small");

template <typename T>
int slot(wfa_ctx* c, int s, int64_t n, T** out) {
    int rc = c->ht[s].ensure((size_t)(n > 0 ? n : 1) * sizeof(T));
    if (rc) return rc;
    *out = c->ht[s].as<T>();
    return WFA_OK;
}

template <typename T>
int upload(wfa_ctx* c, int s, const T* host, int64_t n, T** dev) {
    int rc = slot<T>(c, s, n, dev);
    if (rc) return rc;
    if (n > 0) WFA_HIP_CHECK(hipMemcpyAsync(*dev, host, (size_t)n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return WFA_OK;
}

__global__ void k_iota(int64_t n, int64_t* p) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) p[i] = i;
}
__global__ void k_gather_u64(int64_t n, const uint64_t* __restrict__ src, const int64_t* __restrict__ perm,
                             uint64_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

// min / max of up to 8 key columns in one pass (block reduce, then 64-bit atomics): mm[2k] = min, mm[2k + 1] = max
constexpr int kMaxSortKeys = 8;
struct KeyCols {
    const uint64_t* k[kMaxSortKeys];
    int n;
};
__global__ __launch_bounds__(kTB) void k_key_ranges(int64_t n, KeyCols kc, unsigned long long* __restrict__ mm) {
    __shared__ unsigned long long s_min[kTB / 64], s_max[kTB / 64];
    for (int q = 0; q < kc.n; ++q) {
        unsigned long long lo = ~0ull, hi = 0ull;
        for (int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTB) {
            const unsigned long long v = kc.k[q][i];
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned long long ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
            lo = ol < lo ? ol : lo;
            hi = oh > hi ? oh : hi;
        }
        if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = lo; s_max[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kTB / 64; ++w) { lo = s_min[w] < lo ? s_min[w] : lo; hi = s_max[w] > hi ? s_max[w] : hi; }
            atomicMin(&mm[2 * q], lo);
            atomicMax(&mm[2 * q + 1], hi);
        }
        __syncthreads();
    }
}
__global__ void k_gather_rebased(int64_t n, const uint64_t* __restrict__ src, const int64_t* __restrict__ perm,
                                 uint64_t base, uint64_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]] - base;
}

// perm <- the stable lexicographic order of keys[0] (primary), keys[1], ... ; returns the buffer holding it.
// Every key is sorted on (key - min) over the bits its range needs: a radix pass costs the same whatever the values
// are, and most keys are narrow (dt, board / channel, pid: a few bits or constant; timestamps in ps: ~45 bits).  A
// constant key is skipped.  One extra pass over the keys and one host round trip buy back a third to two thirds of
// the radix passes.
// init_perm (one of this function's own result buffers, or null = the identity): the order the stable passes start from --
// sorting an earlier result by a few more significant keys costs only those keys' passes.
int lexsort(wfa_ctx* c, int64_t n, const uint64_t* const* keys, int n_keys, int64_t** perm_out,
            const int64_t* init_perm = nullptr) {
    int rc;
    int64_t *p0, *p1;
    uint64_t *k0, *k1;
    if (n_keys > kMaxSortKeys) return fail(WFA_E_INVALID, "too many sort keys");
    if ((rc = slot<int64_t>(c, S_PERM0, n, &p0)) || (rc = slot<int64_t>(c, S_PERM1, n, &p1))) return rc;
    if ((rc = slot<uint64_t>(c, S_KTMP0, n, &k0)) || (rc = slot<uint64_t>(c, S_KTMP1, n, &k1))) return rc;
    if (n > 0x7fffffff) return fail(WFA_E_LIMIT, "hit table has %lld rows; the device sort handles < 2^31", (long long)n);
    unsigned long long* d_mm;
    if ((rc = slot<unsigned long long>(c, S_CNT, 2 * kMaxSortKeys, &d_mm))) return rc;
    unsigned long long mm[2 * kMaxSortKeys];
    for (int k = 0; k < kMaxSortKeys; ++k) { mm[2 * k] = ~0ull; mm[2 * k + 1] = 0ull; }
    WFA_HIP_CHECK(hipMemcpyAsync(d_mm, mm, sizeof(mm), hipMemcpyHostToDevice, c->stream));
    KeyCols kc{};
    kc.n = n_keys;
    for (int k = 0; k < n_keys; ++k) kc.k[k] = keys[k];
    const unsigned rb = blocks_for(n) < 1024u ? blocks_for(n) : 1024u;
    hipLaunchKernelGGL(k_key_ranges, dim3(rb), dim3(kTB), 0, c->stream, n, kc, d_mm);
    if (init_perm == p1) { int64_t* t = p0; p0 = p1; p1 = t; }  // the starting order already sits in a result buffer
    else if (init_perm && init_perm != p0) return fail(WFA_E_INVALID, "lexsort: init_perm must be an earlier result");
    if (!init_perm) hipLaunchKernelGGL(k_iota, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, p0);
    WFA_HIP_CHECK(hipMemcpyAsync(mm, d_mm, sizeof(mm), hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    rocprim::double_buffer<uint64_t> kb(k0, k1);
    rocprim::double_buffer<int64_t> pb(p0, p1);
    size_t tmp_bytes = 0;
    WFA_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kb, pb, (size_t)n, 0u, 64u, c->stream));
    if ((rc = c->ht[S_CUB].ensure(tmp_bytes))) return rc;
    for (int k = n_keys - 1; k >= 0; --k) {
        const unsigned long long span = mm[2 * k + 1] - mm[2 * k];
        if (span == 0) continue;  // constant key: the order does not change
        const int bits = 64 - __builtin_clzll(span);
        hipLaunchKernelGGL(k_gather_rebased, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, keys[k], pb.current(),
                           (uint64_t)mm[2 * k], kb.current());
        size_t tb = c->ht[S_CUB].cap;
        WFA_HIP_CHECK(rocprim::radix_sort_pairs(c->ht[S_CUB].ptr, tb, kb, pb, (size_t)n, 0u, (unsigned)bits, c->stream));
    }
    WFA_HIP_CHECK(hipGetLastError());
    *perm_out = pb.current();
    return WFA_OK;
}

struct MaxF64 {
    __device__ double operator()(double a, double b) const { return b > a ? b : a; }
};

// ---- shared prep: absolute windows and sort keys ------------------------------------------------------------
struct HitCols {
    const int64_t* ts;
    const int64_t* pos;
    const int32_t* s;
    const int32_t* e;
    const int32_t* dt;
    const int16_t* board;
    const int16_t* chan;
    const int64_t* rid;
};

// abs = float(timestamp) + (float(edge) - float(position)) * (float(dt) * 1e3)
// (hit_merge.py:75-82, event_grouping.py:365-367; no contraction: the build uses -ffp-contract=off)
// Sort keys.  The window starts are float64 in the reference, but they are integers (picoseconds) whenever timestamps and
// dt are: then the key is that integer -- lexsort sorts (key - min) over the bits the range needs, 44 bits for 12 s of data
// where the float64 bit pattern needs 56 -- and the timestamp key, which only ever breaks ties between EQUAL starts, is
// (timestamp - start): a few thousand samples of range instead of the run's.  A start that is not an integer below 2^62
// raises *inexact: the caller rewrites the two keys in their float64 / plain forms (k_float_keys).
__global__ void k_hit_prep(int64_t n, HitCols h, const double* __restrict__ fix0, const double* __restrict__ fix1,
                           double* __restrict__ abs0, double* __restrict__ abs1,
                           uint64_t* __restrict__ k_abs0, uint64_t* __restrict__ k_dt, uint64_t* __restrict__ k_ts,
                           uint64_t* __restrict__ k_rid, uint64_t* __restrict__ k_chan, int* __restrict__ inexact) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= n) return;
    const double t = (double)h.ts[i], p = (double)h.pos[i], dps = (double)h.dt[i] * 1e3;
    double a0 = t + ((double)h.s[i] - p) * dps, a1 = t + ((double)h.e[i] - p) * dps;
    if (fix0 && fix0[i] == fix0[i]) a0 = fix0[i];
    if (fix1 && fix1[i] == fix1[i]) a1 = fix1[i];
    abs0[i] = a0;
    abs1[i] = a1;
    const bool small = fabs(a0) < 4.0e18;
    const int64_t ai = small ? (int64_t)a0 : 0;
    if (!small || (double)ai != a0) atomicOr(inexact, 1);
    k_abs0[i] = ord_i64(ai);
    if (k_dt) k_dt[i] = (uint64_t)(uint32_t)h.dt[i];
    if (k_ts) k_ts[i] = ord_i64(h.ts[i] - ai);
    if (k_rid) k_rid[i] = ord_i64(h.rid[i]);
    // (board, channel[, dt]) ascending as signed integers
    const uint64_t bc = ((uint64_t)(uint16_t)(h.board[i] ^ (int16_t)0x8000) << 48) |
                        ((uint64_t)(uint16_t)(h.chan[i] ^ (int16_t)0x8000) << 32);
    k_chan[i] = k_dt ? (bc | (uint64_t)(uint32_t)h.dt[i]) : bc;
}

__global__ void k_float_keys(int64_t n, const double* __restrict__ abs0, const int64_t* __restrict__ ts,
                             uint64_t* __restrict__ k_abs0, uint64_t* __restrict__ k_ts) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= n) return;
    k_abs0[i] = ord_f64(abs0[i]);
    if (k_ts) k_ts[i] = ord_i64(ts[i]);
}

// k_hit_prep + the check of its integer keys (one small device -> host copy)
int hit_prep(wfa_ctx* c, int64_t n, const HitCols& h, const double* fix0, const double* fix1, double* abs0, double* abs1,
             uint64_t* k_abs, uint64_t* k_dt, uint64_t* k_ts, uint64_t* k_rid, uint64_t* k_chan) {
    int rc;
    int* d_flag;
    if ((rc = slot<int>(c, S_CNT2, 4, &d_flag))) return rc;
    WFA_HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_hit_prep, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, h, fix0, fix1, abs0, abs1, k_abs, k_dt, k_ts,
                       k_rid, k_chan, d_flag);
    int inexact = 0;
    WFA_HIP_CHECK(hipMemcpyAsync(&inexact, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (inexact) hipLaunchKernelGGL(k_float_keys, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, abs0, h.ts, k_abs, k_ts);
    return WFA_OK;
}

int upload_cols(wfa_ctx* c, int64_t n, const int64_t* ts, const int64_t* pos, const int32_t* s, const int32_t* e,
                const int32_t* dt, const int16_t* board, const int16_t* chan, const int64_t* rid, HitCols* h) {
    int rc;
    int64_t *d_ts, *d_pos, *d_rid;
    int32_t *d_s, *d_e, *d_dt;
    int16_t *d_b, *d_c;
    if ((rc = upload(c, S_TS, ts, n, &d_ts)) || (rc = upload(c, S_POS, pos, n, &d_pos)) ||
        (rc = upload(c, S_START, s, n, &d_s)) || (rc = upload(c, S_END, e, n, &d_e)) ||
        (rc = upload(c, S_DT, dt, n, &d_dt)) || (rc = upload(c, S_BOARD, board, n, &d_b)) ||
        (rc = upload(c, S_CHAN, chan, n, &d_c)) || (rc = upload(c, S_RID, rid, n, &d_rid)))
        return rc;
    *h = HitCols{d_ts, d_pos, d_s, d_e, d_dt, d_b, d_c, d_rid};
    return WFA_OK;
}

// Columns out of device-resident THRESHOLD_HIT_DTYPE rows (60 B packed: position i8 @0, edge_start i4 @16, edge_end i4
// @20, dt i4 @28, timestamp i8 @40, board i2 @48, channel i2 @50, record_id i8 @52; cpu/hit_finder.py:33-49): the rows
// of the last hit pass or of the last RCCL gather feed the hit-table stages without a host round trip.
__global__ void k_unpack_hit_rows(int64_t n, const uint8_t* __restrict__ rows, int64_t* __restrict__ ts,
                                  int64_t* __restrict__ pos, int32_t* __restrict__ s, int32_t* __restrict__ e,
                                  int32_t* __restrict__ dt, int16_t* __restrict__ board, int16_t* __restrict__ chan,
                                  int64_t* __restrict__ rid) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= n) return;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(rows + i * 60);
    pos[i] = (int64_t)(((uint64_t)w[1] << 32) | w[0]);
    s[i] = (int32_t)w[4];
    e[i] = (int32_t)w[5];
    dt[i] = (int32_t)w[7];
    ts[i] = (int64_t)(((uint64_t)w[11] << 32) | w[10]);
    board[i] = (int16_t)(w[12] & 0xffffu);
    chan[i] = (int16_t)(w[12] >> 16);
    rid[i] = (int64_t)(((uint64_t)w[14] << 32) | w[13]);
}

// resident rows selected by wfa_hit_rows_source: 1 = rows of the last hit pass, 2 = rows of the last RCCL gather
int resident_cols(wfa_ctx* c, int64_t n, HitCols* h) {
    const uint8_t* rows = nullptr;
    int64_t have = -1;
    if (c->ht_src == 2) { rows = c->gathered.as<uint8_t>(); have = c->gathered_n; }
    else { rows = c->hit_out.as<uint8_t>(); have = c->pending ? -1 : c->n_hits; }
    if (have < 0) return fail(WFA_E_STATE, "no device-resident hit rows (run a hit pass / a gather first)");
    if (n != have) return fail(WFA_E_INVALID, "the resident hit table has %lld rows, the caller expects %lld", (long long)have, (long long)n);
    int rc;
    int64_t *d_ts, *d_pos, *d_rid;
    int32_t *d_s, *d_e, *d_dt;
    int16_t *d_b, *d_c;
    if ((rc = slot(c, S_TS, n, &d_ts)) || (rc = slot(c, S_POS, n, &d_pos)) || (rc = slot(c, S_START, n, &d_s)) ||
        (rc = slot(c, S_END, n, &d_e)) || (rc = slot(c, S_DT, n, &d_dt)) || (rc = slot(c, S_BOARD, n, &d_b)) ||
        (rc = slot(c, S_CHAN, n, &d_c)) || (rc = slot(c, S_RID, n, &d_rid)))
        return rc;
    hipLaunchKernelGGL(k_unpack_hit_rows, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, rows, d_ts, d_pos, d_s, d_e, d_dt,
                       d_b, d_c, d_rid);
    *h = HitCols{d_ts, d_pos, d_s, d_e, d_dt, d_b, d_c, d_rid};
    return WFA_OK;
}

// ---- event grouping ---------------------------------------------------------------------------------------
__global__ void k_gather_f64(int64_t n, const double* __restrict__ src, const int64_t* __restrict__ perm,
                             double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

// a hit opens a new event <=> its start lies more than `gap` after the latest end seen so far (:457-470)
__global__ void k_event_flags(int64_t n, const double* __restrict__ abs0, const int64_t* __restrict__ perm,
                              const double* __restrict__ run_max, double gap_ps, int64_t* __restrict__ flag) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    flag[j] = (j == 0 || abs0[perm[j]] > run_max[j - 1] + gap_ps) ? 1 : 0;
}

__global__ void k_event_keys(int64_t n, const int64_t* __restrict__ perm, const int64_t* __restrict__ incl,
                             uint64_t* __restrict__ k_event) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j < n) k_event[perm[j]] = (uint64_t)(incl[j] - 1);
}

__global__ void k_event_starts(int64_t n, const uint64_t* __restrict__ k_event, const int64_t* __restrict__ perm,
                               int64_t n_events, int64_t* __restrict__ event_start) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    const uint64_t ev = k_event[perm[j]];
    if (j == 0 || k_event[perm[j - 1]] != ev) event_start[ev] = j;
    if (j == n - 1) event_start[n_events] = n;
}

__global__ void k_event_minmax(int64_t n_events, const int64_t* __restrict__ event_start,
                               const int64_t* __restrict__ perm, const double* __restrict__ abs0,
                               const double* __restrict__ abs1, int64_t* __restrict__ t_min,
                               int64_t* __restrict__ t_max) {
    const int64_t ev = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (ev >= n_events) return;
    const int64_t a = event_start[ev], b = event_start[ev + 1];
    double lo = abs0[perm[a]], hi = abs1[perm[a]];
    for (int64_t j = a + 1; j < b; ++j) {
        const double s = abs0[perm[j]], e = abs1[perm[j]];
        lo = s < lo ? s : lo;
        hi = e > hi ? e : hi;
    }
    t_min[ev] = (int64_t)lo;  // int(np.min(...)): truncation
    t_max[ev] = (int64_t)hi;
}

// ---- hit merge --------------------------------------------------------------------------------------------
// ---- legacy fixed-window grouping (group_multi_channel_hits, event_grouping.py:98-283, 475-525) ----------------------
// After a stable sort by timestamp a cluster takes every hit within `window` of its FIRST hit: the boundaries are the
// chain 0 -> next[0] -> next[next[0]] -> ... with next[i] = upper_bound(ts, ts[i] + window), compared in float64 like
// numpy's searchsorted of an int64 column against a float64 needle.  The chain is marked by pointer jumping: with
// J_k = next^(2^k), after the levels K-1 .. k the marks are { next^m(0) : 2^k | m }; a level is one launch over all hits.
__global__ void k_mc_keys(int64_t n, const int64_t* __restrict__ ts, const int64_t* __restrict__ ch,
                          uint64_t* __restrict__ k_ts, uint64_t* __restrict__ k_ch) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) { k_ts[i] = ord_i64(ts[i]); k_ch[i] = ord_i64(ch[i]); }
}
__global__ void k_mc_sorted_ts(int64_t n, const int64_t* __restrict__ ts, const int64_t* __restrict__ perm,
                               double* __restrict__ ts_f) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n) ts_f[i] = (double)ts[perm[i]];
}
__global__ void k_mc_next(int64_t n, const double* __restrict__ ts_f, double window_ps, int32_t* __restrict__ nxt) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i > n) return;
    if (i == n) { nxt[n] = (int32_t)n; return; }
    const double needle = ts_f[i] + window_ps;
    int64_t lo = i + 1, hi = n;  // ts_f[i] <= needle (window >= 0, or NaN: then nothing is <= needle and the cluster is the hit alone)
    if (!(ts_f[i] <= needle)) { nxt[i] = (int32_t)(i + 1); return; }
    while (lo < hi) {  // first j in (i, n] with ts_f[j] > needle
        const int64_t mid = (lo + hi) >> 1;
        if (ts_f[mid] <= needle) lo = mid + 1; else hi = mid;
    }
    nxt[i] = (int32_t)lo;
}
__global__ void k_mc_jump(int64_t n, const int32_t* __restrict__ j_in, int32_t* __restrict__ j_out) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i <= n) j_out[i] = j_in[j_in[i]];
}
__global__ void k_mc_mark(int64_t n, const int32_t* __restrict__ j_k, int64_t* __restrict__ mark) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n && mark[i]) {
        const int32_t t = j_k[i];
        if (t < n) mark[t] = 1;  // (a mark set during this launch belongs to the level's result as well)
    }
}
__global__ void k_mc_bounds(int64_t n, const int64_t* __restrict__ mark, const int64_t* __restrict__ incl, int64_t n_events,
                            int64_t* __restrict__ bounds) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i < n && mark[i]) bounds[incl[i] - 1] = i;
    if (i == n - 1) bounds[n_events] = n;
}

__global__ void k_merge_gather(int64_t n, const int64_t* __restrict__ perm, const double* __restrict__ abs0,
                               const double* __restrict__ abs1, const int32_t* __restrict__ dt,
                               const uint64_t* __restrict__ k_chan, double* __restrict__ s0, double* __restrict__ s1,
                               int32_t* __restrict__ sdt, uint64_t* __restrict__ sg) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    const int64_t i = perm[j];
    s0[j] = abs0[i]; s1[j] = abs1[i]; sdt[j] = dt[i]; sg[j] = k_chan[i];
}

// Running maximum of abs_end inside a hardware channel (segmented inclusive max-scan): a hit whose start lies
// more than merge_gap after EVERY earlier end of its channel certainly opens a new cluster, because the current
// cluster's end is one of those ends.  Such certain breaks (and channel / dt changes) cut the table into
// segments that chain independently; the total-width cap, which makes the chain a sequential greedy
// segmentation, only ever acts inside a segment.
struct SegMax {
    double v;
    int32_t head;  // 1: a channel starts here
    int32_t pad;
};
struct SegMaxOp {
    __device__ SegMax operator()(const SegMax& a, const SegMax& b) const {
        SegMax r;
        r.head = a.head | b.head;
        r.v = b.head ? b.v : (b.v > a.v ? b.v : a.v);
        r.pad = 0;
        return r;
    }
};
__global__ void k_merge_segin(int64_t n, const double* __restrict__ s1, const uint64_t* __restrict__ sg,
                              SegMax* __restrict__ in) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    SegMax x;
    x.v = s1[j];
    x.head = (j == 0 || sg[j] != sg[j - 1]) ? 1 : 0;
    x.pad = 0;
    in[j] = x;
}

// the chain of hit_merge.py:151-179 over one segment per lane: every lane looks at its own sorted position and
// walks only if a segment starts there (no list of heads: a single atomic counter serialises at ~90 appends / us)
__global__ void k_merge_chain(int64_t n, const double* __restrict__ s0, const double* __restrict__ s1,
                              const int32_t* __restrict__ sdt, const uint64_t* __restrict__ sg,
                              const SegMax* __restrict__ run, int do_merge,
                              double gap_ps, double max_width_ps, int64_t* __restrict__ flag) {
    int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    const bool head = j == 0 || sg[j] != sg[j - 1] || !do_merge || sdt[j] != sdt[j - 1] ||
                      !(s0[j] - run[j - 1].v <= gap_ps);
    if (!head) return;
    const uint64_t grp = sg[j];
    double c_start = s0[j], c_end = s1[j];
    int32_t prev_dt = sdt[j];
    flag[j] = 1;
    for (++j; j < n && sg[j] == grp; ++j) {
        if (!do_merge || sdt[j] != prev_dt || !(s0[j] - run[j - 1].v <= gap_ps)) break;  // the next segment's head
        const double a = s0[j], e = s1[j];
        const double gap = a - c_end;
        const double next_end = e > c_end ? e : c_end;
        const double total = next_end - c_start;
        const bool same_dt = sdt[j] == prev_dt;
        if (do_merge && same_dt && gap <= gap_ps && total <= max_width_ps) {
            flag[j] = 0;
            c_end = next_end;
        } else {
            flag[j] = 1;
            c_start = a;
            c_end = e;
        }
        prev_dt = sdt[j];
    }
}

__global__ void k_cluster_offsets(int64_t n, const int64_t* __restrict__ flag, const int64_t* __restrict__ incl,
                                  int64_t n_clusters, int64_t* __restrict__ offset) {
    const int64_t j = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (j >= n) return;
    if (flag[j]) offset[incl[j] - 1] = j;
    if (j == n - 1) offset[n_clusters] = n;
}

// _emit_cluster (hit_merge.py:256-322) for every cluster; singles are flagged and copied on the host
__global__ void k_merge_emit(int64_t n_clusters, const int64_t* __restrict__ offset, const int64_t* __restrict__ perm,
                             HitCols h, const float* __restrict__ height, const float* __restrict__ integral,
                             int64_t* __restrict__ anchor, float* __restrict__ out_h, float* __restrict__ out_int,
                             int32_t* __restrict__ out_s, int32_t* __restrict__ out_e, float* __restrict__ out_w) {
    __shared__ double s_pw[kPairwiseLevels][kTB];
    const int64_t cl = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (cl >= n_clusters) return;
    const int64_t a = offset[cl], b = offset[cl + 1];
    int64_t best = perm[a];
    double max_h = (double)height[best];
    for (int64_t j = a + 1; j < b; ++j) {
        const double v = (double)height[perm[j]];
        if (v > max_h) max_h = v;  // np.max
    }
    // anchor: the member with the maximum height; ties -> the smallest timestamp, first such member
    bool have = false;
    for (int64_t j = a; j < b; ++j) {
        const int64_t i = perm[j];
        if ((double)height[i] != max_h) continue;
        if (!have || h.ts[i] < h.ts[best]) { best = i; have = true; }
    }
    int32_t smin = h.s[perm[a]], emax = h.e[perm[a]];
    bool one_record = true;
    const int64_t rid0 = h.rid[perm[a]];
    for (int64_t j = a + 1; j < b; ++j) {
        const int64_t i = perm[j];
        smin = h.s[i] < smin ? h.s[i] : smin;
        emax = h.e[i] > emax ? h.e[i] : emax;
        one_record = one_record && h.rid[i] == rid0;
    }
    if (!one_record) { smin = -1; emax = -1; }
    double w = (double)emax - (double)smin;  // python ints; max(.., 0.0)
    w = w > 0.0 ? w : 0.0;
    if (smin < 0 || emax < 0) w = -1.0;
    const double total = np_pairwise_sum([&](int q) { return (double)integral[perm[a + q]]; }, 0, (int)(b - a),
                                         &s_pw[0][threadIdx.x], kTB);
    anchor[cl] = best;
    out_h[cl] = (float)max_h;
    out_int[cl] = (float)total;
    out_s[cl] = smin;
    out_e[cl] = emax;
    out_w[cl] = (float)w;
}

// ---- records builder: global order + pool packing (records_builder.py:115-120, 164-209, 869-945) --------------
__global__ void k_record_keys(int64_t n, const int64_t* __restrict__ ts, const int32_t* __restrict__ pid,
                              const int16_t* __restrict__ board, const int16_t* __restrict__ chan,
                              uint64_t* __restrict__ k_ts, uint64_t* __restrict__ k_pid, uint64_t* __restrict__ k_bc) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= n) return;
    k_ts[i] = ord_i64(ts[i]);
    k_pid[i] = (uint64_t)((uint32_t)pid[i] ^ 0x80000000u);
    k_bc[i] = ((uint64_t)(uint16_t)(board[i] ^ (int16_t)0x8000) << 16) | (uint64_t)(uint16_t)(chan[i] ^ (int16_t)0x8000);
}

// one workgroup per record; 16-byte moves when source and destination are co-aligned, 2-byte moves otherwise
__global__ __launch_bounds__(128) void k_pool_gather(int64_t n, const int64_t* __restrict__ src_off,
                                                     const int64_t* __restrict__ dst_off,
                                                     const int32_t* __restrict__ length,
                                                     const uint16_t* __restrict__ src, uint16_t* __restrict__ dst) {
    const int64_t r = blockIdx.x;
    if (r >= n) return;
    const int64_t so = src_off[r], d0 = dst_off[r];
    const int len = length[r];
    if (len <= 0) return;
    const uint16_t* s = src + so;
    uint16_t* d = dst + d0;
    if (((so | d0) & 7) == 0) {
        const int nv = len >> 3;
        const uint4* s4 = reinterpret_cast<const uint4*>(s);
        uint4* d4 = reinterpret_cast<uint4*>(d);
        for (int i = threadIdx.x; i < nv; i += 128) d4[i] = s4[i];
        for (int i = (nv << 3) + threadIdx.x; i < len; i += 128) d[i] = s[i];
    } else {
        for (int i = threadIdx.x; i < len; i += 128) d[i] = s[i];
    }
}


// ---- K15: delimiter-separated integer text -> int64 columns + uint16 samples (CAEN VX2730 CSV) ------------------
// (utils/formats/vx2730.py:193-340: every reader backend yields the same integers; records_builder.py:212-302 then
// uses columns board / channel / timestamp and the samples from `samples_start` to the end of the row.)
// A row is the bytes between two '\n' (a trailing '\r' dropped); an empty row has 0 fields.  One wave decodes one
// row in 1-KiB tiles: the tile (+ 32 bytes of lookahead) is staged in LDS with 16-byte loads, every lane counts the
// delimiters of its 16 bytes, a wave prefix sum turns that into the field index, and the lane parses the fields that
// FOLLOW its delimiters (a decimal integer is at most 20 characters, so the lookahead always covers it).
constexpr int kCsvTile = 1024;
constexpr int kCsvLook = 32;
constexpr int kCsvMaxMeta = 8;
constexpr int kCsvErrSyntax = 1, kCsvErrRange = 2;

// newline positions in text order: a thread owns 16 bytes; pass 1 counts per block, pass 2 (after a scan of the block
// counts) writes the positions.  (A DeviceSelect over one item per byte took 3.3 ms per 150 MB; this takes 0.1 ms.)
constexpr int kCsvNlBlock = 256;

template <bool FILL>
__global__ __launch_bounds__(kCsvNlBlock) void k_csv_newlines(int64_t n_bytes, const uint8_t* __restrict__ text,
                                                             const int64_t* __restrict__ block_start,
                                                             int64_t* __restrict__ block_count, int64_t* __restrict__ nl) {
    using Scan = rocprim::block_scan<int, kCsvNlBlock>;
    __shared__ typename Scan::storage_type tmp;
    const int64_t a = ((int64_t)blockIdx.x * kCsvNlBlock + threadIdx.x) * 16;
    uint32_t m = 0;
    if (a < n_bytes) {  // the buffer is padded, the clip drops the padding
        const uint4 w = *reinterpret_cast<const uint4*>(text + a);
        const uint32_t v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) m |= (uint32_t)(((v[k >> 2] >> ((k & 3) * 8)) & 0xffu) == '\n') << k;
        if (a + 16 > n_bytes) m &= (1u << (int)(n_bytes - a)) - 1u;
    }
    int before = 0, total = 0;
    Scan().exclusive_scan(__popc(m), before, 0, total, tmp, rocprim::plus<int>());
    if (!FILL) {
        if (threadIdx.x == 0) block_count[blockIdx.x] = total;
        return;
    }
    int64_t o = block_start[blockIdx.x] + before;
    while (m) {
        nl[o++] = a + (__ffs(m) - 1);
        m &= m - 1;
    }
}

__global__ void k_csv_lines(int64_t n_lines, int64_t n_nl, int64_t n_bytes, const uint8_t* __restrict__ text,
                            const int64_t* __restrict__ nl, int64_t* __restrict__ row_start, int64_t* __restrict__ row_end) {
    const int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x;
    if (i >= n_lines) return;
    const int64_t s = i == 0 ? 0 : nl[i - 1] + 1;
    int64_t e = i < n_nl ? nl[i] : n_bytes;
    if (e > s && text[e - 1] == '\r') --e;
    row_start[i] = s;
    row_end[i] = e;
}

__device__ __forceinline__ uint32_t csv_delim_mask(const uint4& w, uint8_t delim) {
    const uint32_t v[4] = {w.x, w.y, w.z, w.w};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) m |= (uint32_t)(((v[k >> 2] >> ((k & 3) * 8)) & 0xffu) == delim) << k;
    return m;
}

// restrict a 16-bit byte mask of the chunk at absolute offset `abs0` to the bytes in [lo, hi)
__device__ __forceinline__ uint32_t csv_clip(uint32_t m, int64_t abs0, int64_t lo, int64_t hi) {
    if (abs0 + 16 <= lo || abs0 >= hi) return 0;
    if (abs0 < lo) m &= ~0u << (int)(lo - abs0);
    if (abs0 + 16 > hi) m &= (1u << (int)(hi - abs0)) - 1u;
    return m;
}

// fields per row and samples per row (fields from samples_start on); one wave per row
__global__ __launch_bounds__(64) void k_csv_count(int64_t n_rows, const uint8_t* __restrict__ text,
                                                  const int64_t* __restrict__ row_start, const int64_t* __restrict__ row_end,
                                                  uint8_t delim, int32_t samples_start, int32_t* __restrict__ n_fields,
                                                  int64_t* __restrict__ n_samples) {
    const int64_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int lane = threadIdx.x;
    const int64_t lo = row_start[r], hi = row_end[r];
    int cnt = 0;
    for (int64_t a = (lo & ~15ll) + lane * 16; a < hi; a += 64 * 16) {
        const uint4 w = *reinterpret_cast<const uint4*>(text + a);
        cnt += __popc(csv_clip(csv_delim_mask(w, delim), a, lo, hi));
    }
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0) {
        const int nf = hi > lo ? cnt + 1 : 0;
        n_fields[r] = nf;
        n_samples[r] = nf > samples_start ? nf - samples_start : 0;
    }
}

struct CsvCols {
    int32_t n_meta;
    int32_t col[kCsvMaxMeta];
};

__global__ __launch_bounds__(64) void k_csv_decode(int64_t n_rows, const uint8_t* __restrict__ text,
                                                   const int64_t* __restrict__ row_start, const int64_t* __restrict__ row_end,
                                                   uint8_t delim, int32_t samples_start, CsvCols cols,
                                                   const int64_t* __restrict__ sample_offset, int64_t* __restrict__ meta,
                                                   uint16_t* __restrict__ samples, unsigned long long* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[kCsvTile + kCsvLook];
    const int64_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int lane = threadIdx.x;
    const int64_t lo = row_start[r], hi = row_end[r];
    if (hi <= lo) return;
    const int64_t s_off = sample_offset[r];
    int field_base = 0;  // delimiters of the row before this tile
    for (int64_t t0 = lo & ~15ll; t0 < hi; t0 += kCsvTile) {
        __syncthreads();
        const uint4 w = *reinterpret_cast<const uint4*>(text + t0 + lane * 16);
        *reinterpret_cast<uint4*>(tile + lane * 16) = w;
        if (lane < kCsvLook / 16)
            *reinterpret_cast<uint4*>(tile + kCsvTile + lane * 16) =
                *reinterpret_cast<const uint4*>(text + t0 + kCsvTile + lane * 16);
        __syncthreads();
        const int64_t a = t0 + lane * 16;
        uint32_t m = csv_clip(csv_delim_mask(w, delim), a, lo, hi);
        const int mine = __popc(m);
        int incl = mine;  // inclusive prefix over the wave
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        int f = field_base + incl - mine;  // index of the field that CONTAINS this lane's first byte
        // the lane owning the row's first byte also parses field 0 (a virtual delimiter before the row)
        bool first = lo >= a && lo < a + 16;
        while (first || m) {
            int64_t q;  // absolute offset of the field's first character
            if (first) { q = lo; first = false; }
            else { const int b = __ffs(m) - 1; m &= m - 1; q = a + b + 1; ++f; }
            const int fidx = (q == lo) ? 0 : f;
            int mj = -1;
            for (int j = 0; j < cols.n_meta; ++j) if (cols.col[j] == fidx) mj = j;
            const bool is_sample = fidx >= samples_start;
            if (mj < 0 && !is_sample) continue;
            int p = (int)(q - t0);
            const int p_end = (int)((hi - t0) < (int64_t)(kCsvTile + kCsvLook) ? (hi - t0) : (kCsvTile + kCsvLook));
            bool neg = false, any = false, bad = false;
            unsigned long long v = 0;
            if (p < p_end && (tile[p] == '-' || tile[p] == '+')) { neg = tile[p] == '-'; ++p; }
            int digits = 0;
            for (; p < p_end; ++p) {
                const uint8_t ch = tile[p];
                if (ch == delim) break;
                if (ch < '0' || ch > '9' || ++digits > 19) { bad = true; break; }
                v = v * 10ull + (unsigned long long)(ch - '0');
                any = true;
            }
            if (!any || bad || v > 0x7fffffffffffffffull) {
                atomicMin(err, ((unsigned long long)r << 24) | ((unsigned long long)(fidx & 0x3fffff) << 2) | kCsvErrSyntax);
                continue;
            }
            const int64_t val = neg ? -(int64_t)v : (int64_t)v;
            if (mj >= 0) meta[r * cols.n_meta + mj] = val;
            if (is_sample) {
                if (val < 0 || val > 65535)
                    atomicMin(err, ((unsigned long long)r << 24) | ((unsigned long long)(fidx & 0x3fffff) << 2) | kCsvErrRange);
                else
                    samples[s_off + (fidx - samples_start)] = (uint16_t)val;
            }
        }
        int total = __shfl(incl, 63);
        field_base += total;
    }
}

}  // namespace
}  // namespace wfa

using namespace wfa;

static int use_device_ht(wfa_ctx* c) {
    if (!c) return fail(WFA_E_INVALID, "ctx is null");
    WFA_HIP_CHECK(hipSetDevice(c->device));
    return WFA_OK;
}

extern "C" {

int wfa_hit_rows_source(wfa_ctx* c, int which) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (which != 1 && which != 2) return fail(WFA_E_INVALID, "row source must be 1 (last hit pass) or 2 (last gather)");
    c->ht_src = which;
    return WFA_OK;
}

int wfa_group_hit_windows_count(wfa_ctx* c, int64_t n, const int64_t* timestamp, const int64_t* position,
                                const int32_t* sample_start, const int32_t* sample_end, const int32_t* dt,
                                const int16_t* board, const int16_t* channel, const int64_t* record_id,
                                const double* abs_start_fix, const double* abs_end_fix, double time_window_ns,
                                int64_t* n_events) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (n < 0 || !n_events) return fail(WFA_E_INVALID, "bad arguments");
    if (time_window_ns < 0) return fail(WFA_E_INVALID, "time_window_ns must be >= 0");
    c->ht_n = -1;
    if (n == 0) { c->ht_n = 0; c->ht_groups = 0; c->ht_kind = 1; *n_events = 0; return WFA_OK; }
    if (n > 0x7fffffffLL) return fail(WFA_E_LIMIT, "hit table has %lld rows; the device stages handle < 2^31", (long long)n);
    HitCols h{};
    const bool resident = !timestamp && !position && !sample_start && !sample_end && !dt && !board && !channel && !record_id;
    if (resident) {  // every column NULL: the device-resident rows (wfa_hit_rows_source); sample window = edge_start / edge_end
        if ((rc = resident_cols(c, n, &h))) return rc;
    } else {
        if (!timestamp || !position || !sample_start || !sample_end || !dt || !board || !channel || !record_id)
            return fail(WFA_E_INVALID, "null column");
        if ((rc = upload_cols(c, n, timestamp, position, sample_start, sample_end, dt, board, channel, record_id, &h))) return rc;
    }
    if ((abs_start_fix == nullptr) != (abs_end_fix == nullptr)) return fail(WFA_E_INVALID, "pass both abs_*_fix arrays or neither");
    double *fix0 = nullptr, *fix1 = nullptr;
    if (abs_start_fix && ((rc = upload(c, S_OUT6, abs_start_fix, n, &fix0)) || (rc = upload(c, S_OUT7, abs_end_fix, n, &fix1)))) return rc;
    double *abs0, *abs1, *ends, *run_max;
    uint64_t *k_abs, *k_dt, *k_ts, *k_rid, *k_chan, *k_ev;
    int64_t *flag, *incl;
    if ((rc = slot(c, S_ABS0, n, &abs0)) || (rc = slot(c, S_ABS1, n, &abs1)) || (rc = slot(c, S_K0, n, &k_abs)) ||
        (rc = slot(c, S_K1, n, &k_dt)) || (rc = slot(c, S_K2, n, &k_ts)) || (rc = slot(c, S_K3, n, &k_rid)) ||
        (rc = slot(c, S_K4, n, &k_chan)) || (rc = slot(c, S_F0, n, &ends)) || (rc = slot(c, S_F1, n, &run_max)) ||
        (rc = slot(c, S_FLAG, n, &flag)) || (rc = slot(c, S_ID, n, &incl)))
        return rc;
    LaunchTimer t(c);
    if ((rc = hit_prep(c, n, h, fix0, fix1, abs0, abs1, k_abs, k_dt, k_ts, k_rid, k_chan))) return rc;
    int64_t* perm = nullptr;
    {
        const uint64_t* keys[4] = {k_abs, k_dt, k_ts, k_rid};  // np.lexsort((record_ids, timestamps, dt, abs_starts))
        if ((rc = lexsort(c, n, keys, 4, &perm))) return rc;
    }
    hipLaunchKernelGGL(k_gather_f64, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, abs1, perm, ends);
    size_t tb = 0;
    WFA_HIP_CHECK(rocprim::inclusive_scan(nullptr, tb, ends, run_max, (size_t)n, MaxF64(), c->stream));
    size_t tb2 = 0;
    WFA_HIP_CHECK(rocprim::inclusive_scan(nullptr, tb2, flag, incl, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    if ((rc = c->ht[S_CUB].ensure(tb > tb2 ? tb : tb2))) return rc;
    tb = c->ht[S_CUB].cap;
    WFA_HIP_CHECK(rocprim::inclusive_scan(c->ht[S_CUB].ptr, tb, ends, run_max, (size_t)n, MaxF64(), c->stream));
    hipLaunchKernelGGL(k_event_flags, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, abs0, perm, run_max,
                       time_window_ns * 1e3, flag);
    tb = c->ht[S_CUB].cap;
    WFA_HIP_CHECK(rocprim::inclusive_scan(c->ht[S_CUB].ptr, tb, flag, incl, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    int64_t n_ev = 0;
    WFA_HIP_CHECK(hipMemcpyAsync(&n_ev, incl + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    // event id as the new primary key; k_dt's buffer is free again (k_chan carries dt)
    k_ev = k_dt;
    hipLaunchKernelGGL(k_event_keys, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, perm, incl, k_ev);
    {
        // np.lexsort((record_ids, timestamps, abs_starts, dt, channels, boards, event_id)).  The hits already stand in
        // (abs_start, dt, timestamp, record_id) order: a stable sort of THAT order by (event, board, channel, dt) leaves
        // every tie -- same event, channel and dt -- in (abs_start, timestamp, record_id) order, which is the reference's.
        // Two narrow keys instead of five (17 radix passes -> 4).
        const uint64_t* keys[2] = {k_ev, k_chan};
        if ((rc = lexsort(c, n, keys, 2, &perm, perm))) return rc;
    }
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    int64_t *ev_start, *t_min, *t_max;
    if ((rc = slot(c, S_OUT0, n_ev + 1, &ev_start)) || (rc = slot(c, S_OUT1, n_ev, &t_min)) || (rc = slot(c, S_OUT2, n_ev, &t_max)))
        return rc;
    hipLaunchKernelGGL(k_event_starts, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, k_ev, perm, n_ev, ev_start);
    hipLaunchKernelGGL(k_event_minmax, dim3(blocks_for(n_ev)), dim3(kTB), 0, c->stream, n_ev, ev_start, perm, abs0, abs1, t_min, t_max);
    WFA_HIP_CHECK(hipGetLastError());
    if ((rc = t.end("hit table: group_hit_windows (sort + scans)"))) return rc;
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->ht_n = n; c->ht_groups = n_ev; c->ht_kind = 1; c->ht_perm = perm;
    *n_events = n_ev;
    return WFA_OK;
}

int wfa_group_hit_windows_fill(wfa_ctx* c, int64_t n, int64_t n_events, int64_t* order, int64_t* event_start,
                               int64_t* t_min, int64_t* t_max) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (c->ht_n < 0 || c->ht_kind != 1) return fail(WFA_E_STATE, "no event grouping pass has been run");
    if (n != c->ht_n || n_events != c->ht_groups)
        return fail(WFA_E_INVALID, "caller expects %lld hits / %lld events, the pass produced %lld / %lld", (long long)n,
                    (long long)n_events, (long long)c->ht_n, (long long)c->ht_groups);
    if (!event_start) return fail(WFA_E_INVALID, "event_start is null");
    if (n == 0) { event_start[0] = 0; return WFA_OK; }
    if (!order || !t_min || !t_max) return fail(WFA_E_INVALID, "null output");
    WFA_HIP_CHECK(hipMemcpyAsync(order, c->ht_perm, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(event_start, c->ht[S_OUT0].ptr, (size_t)(n_events + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(t_min, c->ht[S_OUT1].ptr, (size_t)n_events * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(t_max, c->ht[S_OUT2].ptr, (size_t)n_events * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_group_multi_channel_count(wfa_ctx* c, int64_t n, const int64_t* timestamp, const int64_t* channel,
                                  double time_window_ps, int64_t* n_events) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (n < 0 || !n_events) return fail(WFA_E_INVALID, "bad arguments");
    c->ht_n = -1;
    if (n == 0) { c->ht_n = 0; c->ht_groups = 0; c->ht_kind = 3; *n_events = 0; return WFA_OK; }
    if (!timestamp || !channel) return fail(WFA_E_INVALID, "null column");
    if (n >= 0x7fffffffLL) return fail(WFA_E_LIMIT, "hit table has %lld rows; the device stages handle < 2^31 - 1", (long long)n);
    int levels = 1;
    while ((1ll << levels) < n) ++levels;  // next^(2^levels) of any hit is n
    int64_t *ts, *ch, *mark, *incl;
    uint64_t *k_ts, *k_ch, *k_ev;
    double* ts_f;
    int32_t* jump;
    if ((rc = upload(c, S_TS, timestamp, n, &ts)) || (rc = upload(c, S_POS, channel, n, &ch)) ||
        (rc = slot(c, S_K0, n, &k_ts)) || (rc = slot(c, S_K1, n, &k_ch)) || (rc = slot(c, S_K2, n, &k_ev)) ||
        (rc = slot(c, S_ABS0, n, &ts_f)) || (rc = slot(c, S_FLAG, n, &mark)) || (rc = slot(c, S_ID, n, &incl)) ||
        (rc = slot(c, S_OUT3, (int64_t)levels * (n + 1), &jump)))
        return rc;
    LaunchTimer t(c);
    hipLaunchKernelGGL(k_mc_keys, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, ts, ch, k_ts, k_ch);
    int64_t* perm = nullptr;
    {
        const uint64_t* keys[1] = {k_ts};  // df.sort_values("timestamp"), stable
        if ((rc = lexsort(c, n, keys, 1, &perm))) return rc;
    }
    hipLaunchKernelGGL(k_mc_sorted_ts, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, ts, perm, ts_f);
    hipLaunchKernelGGL(k_mc_next, dim3(blocks_for(n + 1)), dim3(kTB), 0, c->stream, n, ts_f, time_window_ps, jump);
    for (int k = 1; k < levels; ++k)
        hipLaunchKernelGGL(k_mc_jump, dim3(blocks_for(n + 1)), dim3(kTB), 0, c->stream, n, jump + (int64_t)(k - 1) * (n + 1),
                           jump + (int64_t)k * (n + 1));
    WFA_HIP_CHECK(hipMemsetAsync(mark, 0, (size_t)n * sizeof(int64_t), c->stream));
    const int64_t one = 1;
    WFA_HIP_CHECK(hipMemcpyAsync(mark, &one, sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    for (int k = levels - 1; k >= 0; --k)
        hipLaunchKernelGGL(k_mc_mark, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, jump + (int64_t)k * (n + 1), mark);
    size_t tb = 0;
    WFA_HIP_CHECK(rocprim::inclusive_scan(nullptr, tb, mark, incl, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    if ((rc = c->ht[S_CUB].ensure(tb))) return rc;
    tb = c->ht[S_CUB].cap;
    WFA_HIP_CHECK(rocprim::inclusive_scan(c->ht[S_CUB].ptr, tb, mark, incl, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    int64_t n_ev = 0;
    WFA_HIP_CHECK(hipMemcpyAsync(&n_ev, incl + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    hipLaunchKernelGGL(k_event_keys, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, perm, incl, k_ev);
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));  // (`one` and n_ev live on this stack frame)
    int64_t* bounds;
    if ((rc = slot(c, S_OUT0, n_ev + 1, &bounds))) return rc;
    hipLaunchKernelGGL(k_mc_bounds, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, mark, incl, n_ev, bounds);
    {
        // inside a cluster by channel, equal channels in timestamp order: np.lexsort((arange, channel, event)) of the sorted
        // table = a stable sort of the timestamp order by (event, channel)
        const uint64_t* keys[2] = {k_ev, k_ch};
        if ((rc = lexsort(c, n, keys, 2, &perm, perm))) return rc;
    }
    WFA_HIP_CHECK(hipGetLastError());
    if ((rc = t.end("hit table: group_multi_channel_hits (sort + pointer jumping)"))) return rc;
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->ht_n = n; c->ht_groups = n_ev; c->ht_kind = 3; c->ht_perm = perm;
    *n_events = n_ev;
    return WFA_OK;
}

int wfa_group_multi_channel_fill(wfa_ctx* c, int64_t n, int64_t n_events, int64_t* order, int64_t* bounds) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (c->ht_n < 0 || c->ht_kind != 3) return fail(WFA_E_STATE, "no multi-channel grouping pass has been run");
    if (n != c->ht_n || n_events != c->ht_groups)
        return fail(WFA_E_INVALID, "caller expects %lld hits / %lld events, the pass produced %lld / %lld", (long long)n,
                    (long long)n_events, (long long)c->ht_n, (long long)c->ht_groups);
    if (!bounds) return fail(WFA_E_INVALID, "bounds is null");
    if (n == 0) { bounds[0] = 0; return WFA_OK; }
    if (!order) return fail(WFA_E_INVALID, "order is null");
    WFA_HIP_CHECK(hipMemcpyAsync(order, c->ht_perm, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(bounds, c->ht[S_OUT0].ptr, (size_t)(n_events + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_hit_merge_count(wfa_ctx* c, int64_t n, const int64_t* timestamp, const int64_t* position,
                        const int32_t* edge_start, const int32_t* edge_end, const int32_t* dt, const int16_t* board,
                        const int16_t* channel, double merge_gap_ns, double max_total_width_ns, int64_t* n_clusters) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (n < 0 || !n_clusters) return fail(WFA_E_INVALID, "bad arguments");
    c->ht_n = -1;
    if (n == 0) { c->ht_n = 0; c->ht_groups = 0; c->ht_kind = 2; *n_clusters = 0; return WFA_OK; }
    if (n > 0x7fffffffLL) return fail(WFA_E_LIMIT, "hit table has %lld rows; the device stages handle < 2^31", (long long)n);
    HitCols h{};
    const bool resident = !timestamp && !position && !edge_start && !edge_end && !dt && !board && !channel;
    if (resident) {  // every column NULL: the device-resident rows (wfa_hit_rows_source)
        if ((rc = resident_cols(c, n, &h))) return rc;
    } else {
        if (!timestamp || !position || !edge_start || !edge_end || !dt || !board || !channel)
            return fail(WFA_E_INVALID, "null column");
        if ((rc = upload_cols(c, n, timestamp, position, edge_start, edge_end, dt, board, channel, timestamp, &h))) return rc;
    }
    double *abs0, *abs1, *s0, *s1;
    uint64_t *k_abs, *k_chan, *sg;
    int32_t* sdt;
    int64_t *flag, *incl;
    if ((rc = slot(c, S_ABS0, n, &abs0)) || (rc = slot(c, S_ABS1, n, &abs1)) || (rc = slot(c, S_K0, n, &k_abs)) ||
        (rc = slot(c, S_K4, n, &k_chan)) || (rc = slot(c, S_F0, n, &s0)) || (rc = slot(c, S_F1, n, &s1)) ||
        (rc = slot(c, S_K1, n, &sdt)) || (rc = slot(c, S_SG, n, &sg)) || (rc = slot(c, S_FLAG, n, &flag)) ||
        (rc = slot(c, S_ID, n, &incl)))
        return rc;
    LaunchTimer t(c);
    if ((rc = hit_prep(c, n, h, nullptr, nullptr, abs0, abs1, k_abs, nullptr, nullptr, nullptr, k_chan))) return rc;
    int64_t* perm = nullptr;
    {
        // per hardware channel (ascending board, channel), stable by abs_start (hit_merge.py:137-149)
        const uint64_t* keys[2] = {k_chan, k_abs};
        if ((rc = lexsort(c, n, keys, 2, &perm))) return rc;
    }
    hipLaunchKernelGGL(k_merge_gather, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, perm, abs0, abs1, h.dt, k_chan, s0, s1, sdt, sg);
    SegMax *seg_in, *seg_run;
    if ((rc = slot(c, S_OUT1, n, &seg_in)) || (rc = slot(c, S_OUT2, n, &seg_run))) return rc;
    const int do_merge = merge_gap_ns > 0 ? 1 : 0;
    const double gap_ps = merge_gap_ns * 1e3;
    hipLaunchKernelGGL(k_merge_segin, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, s1, sg, seg_in);
    size_t tb = 0, tb_seg = 0;
    WFA_HIP_CHECK(rocprim::inclusive_scan(nullptr, tb_seg, seg_in, seg_run, (size_t)n, SegMaxOp(), c->stream));
    WFA_HIP_CHECK(rocprim::inclusive_scan(nullptr, tb, flag, incl, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    if ((rc = c->ht[S_CUB].ensure(tb > tb_seg ? tb : tb_seg))) return rc;
    tb_seg = c->ht[S_CUB].cap;
    WFA_HIP_CHECK(rocprim::inclusive_scan(c->ht[S_CUB].ptr, tb_seg, seg_in, seg_run, (size_t)n, SegMaxOp(), c->stream));
    hipLaunchKernelGGL(k_merge_chain, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, s0, s1, sdt, sg, seg_run, do_merge,
                       gap_ps, max_total_width_ns * 1e3, flag);
    tb = c->ht[S_CUB].cap;
    WFA_HIP_CHECK(rocprim::inclusive_scan(c->ht[S_CUB].ptr, tb, flag, incl, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    int64_t n_cl = 0;
    WFA_HIP_CHECK(hipMemcpyAsync(&n_cl, incl + (n - 1), sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    int64_t* offset;
    if ((rc = slot(c, S_OUT0, n_cl + 1, &offset))) return rc;
    hipLaunchKernelGGL(k_cluster_offsets, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, flag, incl, n_cl, offset);
    WFA_HIP_CHECK(hipGetLastError());
    if ((rc = t.end("hit table: hit_merge clusters (sort + chain)"))) return rc;
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->ht_n = n; c->ht_groups = n_cl; c->ht_kind = 2; c->ht_perm = perm;
    *n_clusters = n_cl;
    return WFA_OK;
}

int wfa_hit_merge_fill(wfa_ctx* c, int64_t n, int64_t n_clusters, int64_t* order, int64_t* cluster_offset) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (c->ht_n < 0 || c->ht_kind != 2) return fail(WFA_E_STATE, "no hit merge pass has been run");
    if (n != c->ht_n || n_clusters != c->ht_groups)
        return fail(WFA_E_INVALID, "caller expects %lld hits / %lld clusters, the pass produced %lld / %lld", (long long)n,
                    (long long)n_clusters, (long long)c->ht_n, (long long)c->ht_groups);
    if (!cluster_offset) return fail(WFA_E_INVALID, "cluster_offset is null");
    if (n == 0) { cluster_offset[0] = 0; return WFA_OK; }
    if (!order) return fail(WFA_E_INVALID, "order is null");
    WFA_HIP_CHECK(hipMemcpyAsync(order, c->ht_perm, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(cluster_offset, c->ht[S_OUT0].ptr, (size_t)(n_clusters + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_hit_merge_emit(wfa_ctx* c, int64_t n, const int64_t* timestamp, const int32_t* sample_start,
                       const int32_t* sample_end, const int64_t* record_id, const float* height, const float* integral,
                       int64_t n_members, const int64_t* member_hit, int64_t n_clusters, const int64_t* cluster_offset,
                       int64_t* anchor, float* out_height, float* out_integral, int32_t* out_start, int32_t* out_end,
                       float* out_width) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (n < 0 || n_members < 0 || n_clusters < 0) return fail(WFA_E_INVALID, "negative size");
    if (n_clusters == 0) return WFA_OK;
    if (!timestamp || !sample_start || !sample_end || !record_id || !height || !integral || !member_hit || !cluster_offset ||
        !anchor || !out_height || !out_integral || !out_start || !out_end || !out_width)
        return fail(WFA_E_INVALID, "null argument");
    // membership must be well-formed before a kernel indexes with it
    if (cluster_offset[0] != 0 || cluster_offset[n_clusters] != n_members) return fail(WFA_E_INVALID, "cluster_offset does not span the members");
    for (int64_t k = 0; k < n_clusters; ++k)
        if (cluster_offset[k + 1] <= cluster_offset[k]) return fail(WFA_E_INVALID, "empty or unordered cluster %lld", (long long)k);
    for (int64_t j = 0; j < n_members; ++j)
        if (member_hit[j] < 0 || member_hit[j] >= n) return fail(WFA_E_INVALID, "hit index %lld out of range", (long long)member_hit[j]);
    HitCols h{};
    int64_t *d_ts, *d_rid, *d_m, *d_off, *d_anchor;
    int32_t *d_s, *d_e, *o_s, *o_e;
    float *d_h, *d_i, *o_h, *o_i, *o_w;
    if ((rc = upload(c, S_TS, timestamp, n, &d_ts)) || (rc = upload(c, S_START, sample_start, n, &d_s)) ||
        (rc = upload(c, S_END, sample_end, n, &d_e)) || (rc = upload(c, S_RID, record_id, n, &d_rid)) ||
        (rc = upload(c, S_HEIGHT, height, n, &d_h)) || (rc = upload(c, S_INTEGRAL, integral, n, &d_i)) ||
        (rc = upload(c, S_PERM0, member_hit, n_members, &d_m)) || (rc = upload(c, S_OUT0, cluster_offset, n_clusters + 1, &d_off)))
        return rc;
    if ((rc = slot(c, S_OUT1, n_clusters, &d_anchor)) || (rc = slot(c, S_OUT2, n_clusters, &o_h)) || (rc = slot(c, S_OUT3, n_clusters, &o_i)) ||
        (rc = slot(c, S_OUT4, n_clusters, &o_s)) || (rc = slot(c, S_OUT5, n_clusters, &o_e)) || (rc = slot(c, S_OUT6, n_clusters, &o_w)))
        return rc;
    h.ts = d_ts; h.s = d_s; h.e = d_e; h.rid = d_rid;
    c->ht_n = -1;
    LaunchTimer t(c);
    hipLaunchKernelGGL(k_merge_emit, dim3(blocks_for(n_clusters)), dim3(kTB), 0, c->stream, n_clusters, d_off, d_m, h, d_h, d_i,
                       d_anchor, o_h, o_i, o_s, o_e, o_w);
    WFA_HIP_CHECK(hipGetLastError());
    if ((rc = t.end("hit table: hit_merge emit"))) return rc;
    const size_t m = (size_t)n_clusters;
    WFA_HIP_CHECK(hipMemcpyAsync(anchor, d_anchor, m * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(out_height, o_h, m * 4, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(out_integral, o_i, m * 4, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(out_start, o_s, m * 4, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(out_end, o_e, m * 4, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(out_width, o_w, m * 4, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_records_sort(wfa_ctx* c, int64_t n, const int64_t* timestamp, const int32_t* pid, const int16_t* board,
                     const int16_t* channel, int64_t* order) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (n < 0) return fail(WFA_E_INVALID, "negative size");
    if (n == 0) return WFA_OK;
    if (!timestamp || !pid || !board || !channel || !order) return fail(WFA_E_INVALID, "null argument");
    int64_t* d_ts;
    int32_t* d_pid;
    int16_t *d_b, *d_c;
    uint64_t *k_ts, *k_pid, *k_bc;
    if ((rc = upload(c, S_TS, timestamp, n, &d_ts)) || (rc = upload(c, S_DT, pid, n, &d_pid)) ||
        (rc = upload(c, S_BOARD, board, n, &d_b)) || (rc = upload(c, S_CHAN, channel, n, &d_c)) ||
        (rc = slot(c, S_K0, n, &k_ts)) || (rc = slot(c, S_K1, n, &k_pid)) || (rc = slot(c, S_K2, n, &k_bc)))
        return rc;
    c->ht_n = -1;
    LaunchTimer t(c);
    hipLaunchKernelGGL(k_record_keys, dim3(blocks_for(n)), dim3(kTB), 0, c->stream, n, d_ts, d_pid, d_b, d_c, k_ts, k_pid, k_bc);
    int64_t* perm = nullptr;
    const uint64_t* keys[3] = {k_ts, k_pid, k_bc};  // np.lexsort((seq, channel, board, pid, timestamp)); seq = stability
    if ((rc = lexsort(c, n, keys, 3, &perm))) return rc;
    if ((rc = t.end("records: global sort order"))) return rc;
    WFA_HIP_CHECK(hipMemcpyAsync(order, perm, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_pool_gather(wfa_ctx* c, int64_t n, const int64_t* src_offset, const int32_t* length, const uint16_t* src_pool,
                    int64_t src_samples, int64_t* out_offset, uint16_t* out_pool, int64_t out_samples) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (n < 0 || src_samples < 0 || out_samples < 0) return fail(WFA_E_INVALID, "negative size");
    if (n > 0 && (!src_offset || !length || !out_offset)) return fail(WFA_E_INVALID, "null argument");
    const bool from_csv = !src_pool && src_samples > 0;  // the samples the last wfa_csv_decode_fill left on the device
    if (from_csv && (!c->csv_filled || c->csv_samples != src_samples))
        return fail(WFA_E_INVALID, "src_pool is null and no decoded CSV samples of that size are resident");
    // every slice is checked before a kernel indexes with it; offsets of the packed pool are the running sum
    int64_t cursor = 0;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t len = length[r] > 0 ? length[r] : 0;
        if (len > 0 && (src_offset[r] < 0 || src_offset[r] + len > src_samples))
            return fail(WFA_E_INVALID, "record %lld: slice [%lld, %lld) outside the source pool of %lld samples", (long long)r,
                        (long long)src_offset[r], (long long)(src_offset[r] + len), (long long)src_samples);
        out_offset[r] = cursor;
        cursor += len;
    }
    if (cursor != out_samples)
        return fail(WFA_E_INVALID, "lengths add up to %lld samples, caller expects %lld", (long long)cursor, (long long)out_samples);
    uint16_t* d_src;
    int64_t *d_so, *d_do;
    int32_t* d_len;
    if (from_csv) d_src = c->ht[S_CSV].as<uint16_t>();
    else if ((rc = upload(c, S_F0, src_pool, src_samples, &d_src))) return rc;
    if ((rc = upload(c, S_K0, src_offset, n, &d_so)) ||
        (rc = upload(c, S_K1, (const int64_t*)out_offset, n, &d_do)) || (rc = upload(c, S_K2, length, n, &d_len)))
        return rc;
    if ((rc = c->pool_u16.ensure((size_t)(out_samples > 0 ? out_samples : 1) * sizeof(uint16_t)))) return rc;
    {
        LaunchTimer t(c);
        if (n > 0)
            hipLaunchKernelGGL(k_pool_gather, dim3((unsigned)n), dim3(128), 0, c->stream, n, d_so, d_do, d_len, d_src,
                               c->pool_u16.as<uint16_t>());
        WFA_HIP_CHECK(hipGetLastError());
        if ((rc = t.end("k_pool_gather"))) return rc;
    }
    if (out_pool && out_samples > 0)
        WFA_HIP_CHECK(hipMemcpyAsync(out_pool, c->pool_u16.ptr, (size_t)out_samples * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    // the packed pool is now the resident wave_pool
    c->pool_n = out_samples;
    c->have_u16 = true;
    c->have_f32 = false;
    c->filter_keep = false;
    c->have_records = false;
    c->shadow_valid = false;
    c->pad_ok = false;
    return WFA_OK;
}

// Host-only: header walk of a CAEN V1725 DAW_DEMO binary stream (utils/formats/v1725.py:66-114).  The stream is a
// chain of variable-length events, so the walk is sequential; it touches 16 + 12 bytes per wave and leaves the
// payloads where they are -- wfa_pool_gather moves them on the GPU.
int wfa_v1725_index(const uint8_t* buf, int64_t n_bytes, int64_t capacity, int16_t* channel, int64_t* timestamp,
                    uint8_t* trunc, uint16_t* baseline, int64_t* payload_offset, int32_t* n_samples, int64_t* n_waves) {
    if (n_bytes < 0 || capacity < 0 || !n_waves || (n_bytes > 0 && !buf)) return fail(WFA_E_INVALID, "bad arguments");
    if (capacity > 0 && (!channel || !timestamp || !trunc || !baseline || !payload_offset || !n_samples))
        return fail(WFA_E_INVALID, "null output column");
    char err[160];
    if (host::v1725_index(buf, n_bytes, capacity, channel, timestamp, trunc, baseline, payload_offset, n_samples, n_waves, err,
                          sizeof(err)))
        return fail(WFA_E_INVALID, "%s", err);
    return WFA_OK;
}

// K15 entry points: see include/wfa_hip.h
int wfa_csv_decode_count(wfa_ctx* c, const uint8_t* text, int64_t n_bytes, int delimiter, int32_t samples_start,
                         int64_t* n_rows, int64_t* n_samples) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (n_bytes < 0 || !n_rows || !n_samples) return fail(WFA_E_INVALID, "bad arguments");
    if (n_bytes > 0 && !text) return fail(WFA_E_INVALID, "text is null");
    if (delimiter <= 0 || delimiter > 127 || delimiter == '\n' || delimiter == '\r' || (delimiter >= '0' && delimiter <= '9') ||
        delimiter == '-' || delimiter == '+')
        return fail(WFA_E_INVALID, "delimiter must be an ASCII character that is not a digit, a sign or a line end");
    if (samples_start < 0) return fail(WFA_E_INVALID, "samples_start must be >= 0");
    if (n_bytes > 0x7fffffff) return fail(WFA_E_LIMIT, "text of %lld bytes; one decode call handles < 2^31 bytes", (long long)n_bytes);
    c->csv_rows = -1; c->csv_samples = -1; c->csv_filled = false;
    *n_rows = 0; *n_samples = 0;
    if (n_bytes == 0) { c->csv_rows = 0; c->csv_samples = 0; c->csv_bytes = 0; return WFA_OK; }
    uint8_t* d_text;
    if ((rc = slot<uint8_t>(c, S_ABS0, n_bytes + 2 * kCsvTile, &d_text))) return rc;
    WFA_HIP_CHECK(hipMemcpyAsync(d_text, text, (size_t)n_bytes, hipMemcpyHostToDevice, c->stream));
    WFA_HIP_CHECK(hipMemsetAsync(d_text + n_bytes, 0, 2 * kCsvTile, c->stream));  // tiles read past the last row
    const int64_t nb = (n_bytes + kCsvNlBlock * 16 - 1) / (kCsvNlBlock * 16);
    int64_t *d_bc, *d_bs;
    if ((rc = slot<int64_t>(c, S_OUT2, nb, &d_bc)) || (rc = slot<int64_t>(c, S_OUT3, nb, &d_bs))) return rc;
    LaunchTimer t(c);
    hipLaunchKernelGGL((k_csv_newlines<false>), dim3((unsigned)nb), dim3(kCsvNlBlock), 0, c->stream, n_bytes, d_text,
                       (const int64_t*)nullptr, d_bc, (int64_t*)nullptr);
    size_t tmp = 0;
    WFA_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp, d_bc, d_bs, (int64_t)0, (size_t)nb, rocprim::plus<int64_t>(), c->stream));
    if ((rc = c->ht[S_CUB].ensure(tmp))) return rc;
    WFA_HIP_CHECK(rocprim::exclusive_scan(c->ht[S_CUB].ptr, tmp, d_bc, d_bs, (int64_t)0, (size_t)nb, rocprim::plus<int64_t>(), c->stream));
    int64_t last[2] = {0, 0};
    WFA_HIP_CHECK(hipMemcpyAsync(&last[0], d_bs + nb - 1, 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(&last[1], d_bc + nb - 1, 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    const int64_t n_nl = last[0] + last[1];
    int64_t* d_nl;
    if ((rc = slot<int64_t>(c, S_K0, n_nl, &d_nl))) return rc;
    hipLaunchKernelGGL((k_csv_newlines<true>), dim3((unsigned)nb), dim3(kCsvNlBlock), 0, c->stream, n_bytes, d_text,
                       (const int64_t*)d_bs, (int64_t*)nullptr, d_nl);
    const int64_t n_lines = n_nl + (text[n_bytes - 1] != '\n' ? 1 : 0);
    int64_t *d_rs, *d_re, *d_ns, *d_so;
    int32_t* d_nf;
    if ((rc = slot<int64_t>(c, S_K1, n_lines, &d_rs)) || (rc = slot<int64_t>(c, S_K2, n_lines, &d_re)) ||
        (rc = slot<int32_t>(c, S_OUT0, n_lines, &d_nf)) || (rc = slot<int64_t>(c, S_K3, n_lines, &d_ns)) ||
        (rc = slot<int64_t>(c, S_K4, n_lines + 1, &d_so)))
        return rc;
    if (n_lines > 0) {
        hipLaunchKernelGGL(k_csv_lines, dim3(blocks_for(n_lines)), dim3(kTB), 0, c->stream, n_lines, n_nl, n_bytes, d_text,
                           d_nl, d_rs, d_re);
        hipLaunchKernelGGL(k_csv_count, dim3((unsigned)n_lines), dim3(64), 0, c->stream, n_lines, d_text, d_rs, d_re,
                           (uint8_t)delimiter, samples_start, d_nf, d_ns);
        size_t tb = 0;
        WFA_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, d_ns, d_so, (int64_t)0, (size_t)n_lines, rocprim::plus<int64_t>(), c->stream));
        if ((rc = c->ht[S_CUB].ensure(tb))) return rc;
        WFA_HIP_CHECK(rocprim::exclusive_scan(c->ht[S_CUB].ptr, tb, d_ns, d_so, (int64_t)0, (size_t)n_lines, rocprim::plus<int64_t>(), c->stream));
    }
    WFA_HIP_CHECK(hipGetLastError());
    if ((rc = t.end("csv: index rows + count fields"))) return rc;
    int64_t last_off = 0, last_n = 0;
    if (n_lines > 0) {
        WFA_HIP_CHECK(hipMemcpyAsync(&last_off, d_so + n_lines - 1, 8, hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipMemcpyAsync(&last_n, d_ns + n_lines - 1, 8, hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    c->csv_rows = n_lines; c->csv_samples = last_off + last_n; c->csv_bytes = n_bytes;
    c->csv_samples_start = samples_start; c->csv_delim = delimiter;
    *n_rows = n_lines; *n_samples = c->csv_samples;
    return WFA_OK;
}

int wfa_csv_decode_fill(wfa_ctx* c, int64_t n_rows, int32_t n_meta, const int32_t* meta_cols, int64_t* meta,
                        int64_t* row_offset, int32_t* n_fields, int64_t* sample_offset, uint16_t* samples,
                        int64_t n_samples) {
    int rc = use_device_ht(c);
    if (rc) return rc;
    if (c->csv_rows < 0) return fail(WFA_E_STATE, "no wfa_csv_decode_count pass has been run");
    if (n_rows != c->csv_rows || n_samples != c->csv_samples)
        return fail(WFA_E_INVALID, "the count pass found %lld rows / %lld samples", (long long)c->csv_rows, (long long)c->csv_samples);
    if (n_meta < 0 || n_meta > kCsvMaxMeta) return fail(WFA_E_INVALID, "n_meta must be 0..%d", kCsvMaxMeta);
    if (n_meta > 0 && (!meta_cols || !meta)) return fail(WFA_E_INVALID, "null meta arguments");
    CsvCols cols{};
    cols.n_meta = n_meta;
    for (int j = 0; j < n_meta; ++j) {
        if (meta_cols[j] < 0 || meta_cols[j] >= c->csv_samples_start)
            return fail(WFA_E_INVALID, "meta column %d is not before samples_start = %d", meta_cols[j], c->csv_samples_start);
        cols.col[j] = meta_cols[j];
    }
    if (n_rows == 0) { c->csv_filled = true; return WFA_OK; }
    int64_t* d_meta;
    uint16_t* d_samples;
    unsigned long long* d_err;
    if ((rc = slot<int64_t>(c, S_OUT1, n_rows * (n_meta > 0 ? n_meta : 1), &d_meta)) ||
        (rc = slot<uint16_t>(c, S_CSV, n_samples, &d_samples)) || (rc = slot<unsigned long long>(c, S_FLAG, 1, &d_err)))
        return rc;
    WFA_HIP_CHECK(hipMemsetAsync(d_err, 0xff, 8, c->stream));
    WFA_HIP_CHECK(hipMemsetAsync(d_meta, 0, (size_t)n_rows * (n_meta > 0 ? n_meta : 1) * 8, c->stream));
    {
        LaunchTimer t(c);
        hipLaunchKernelGGL(k_csv_decode, dim3((unsigned)n_rows), dim3(64), 0, c->stream, n_rows, c->ht[S_ABS0].as<uint8_t>(),
                           c->ht[S_K1].as<int64_t>(), c->ht[S_K2].as<int64_t>(), (uint8_t)c->csv_delim, c->csv_samples_start,
                           cols, c->ht[S_K4].as<int64_t>(), d_meta, d_samples, d_err);
        WFA_HIP_CHECK(hipGetLastError());
        if ((rc = t.end("k_csv_decode"))) return rc;
    }
    unsigned long long err = 0;
    WFA_HIP_CHECK(hipMemcpyAsync(&err, d_err, 8, hipMemcpyDeviceToHost, c->stream));
    if (n_meta > 0) WFA_HIP_CHECK(hipMemcpyAsync(meta, d_meta, (size_t)n_rows * n_meta * 8, hipMemcpyDeviceToHost, c->stream));
    if (row_offset) WFA_HIP_CHECK(hipMemcpyAsync(row_offset, c->ht[S_K1].ptr, (size_t)n_rows * 8, hipMemcpyDeviceToHost, c->stream));
    if (n_fields) WFA_HIP_CHECK(hipMemcpyAsync(n_fields, c->ht[S_OUT0].ptr, (size_t)n_rows * 4, hipMemcpyDeviceToHost, c->stream));
    if (sample_offset) WFA_HIP_CHECK(hipMemcpyAsync(sample_offset, c->ht[S_K4].ptr, (size_t)n_rows * 8, hipMemcpyDeviceToHost, c->stream));
    if (samples && n_samples > 0)
        WFA_HIP_CHECK(hipMemcpyAsync(samples, d_samples, (size_t)n_samples * 2, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (err != ~0ull) {
        const long long row = (long long)(err >> 24), field = (long long)((err >> 2) & 0x3fffff);
        if ((err & 3) == kCsvErrRange)
            return fail(WFA_E_INVALID, "row %lld field %lld: sample outside the uint16 range", row, field);
        return fail(WFA_E_INVALID, "row %lld field %lld: not a decimal integer", row, field);
    }
    c->csv_filled = true;
    return WFA_OK;
}

}  // extern "C"
