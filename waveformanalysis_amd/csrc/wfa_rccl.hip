// RCCL leg of libwfa_hip.so: the one exchange step of the path.
//
// Records shard by hardware channel with no data-path collective; only
// event grouping (reference: core/processing/event_grouping.py:286-471) needs every channel's
// hits in one place.  Payload = 60-72 B rows, a few MB per rank: latency-bound, so this is one
// count all-gather plus one grouped send/recv to the root over xGMI -- no ring tuning needed.

#include <rccl/rccl.h>

#include <utility>

#include "wfa_common.hpp"

using namespace wfa;

#define WFA_NCCL_CHECK(expr)                                                                   \
    do {                                                                                       \
        ncclResult_t _r = (expr);                                                              \
        if (_r != ncclSuccess)                                                                 \
            return ::wfa::fail(WFA_E_RCCL, "%s failed: %s", #expr, ncclGetErrorString(_r));    \
    } while (0)

extern "C" {

int wfa_rccl_unique_id(void* id128) {
    if (!id128) return fail(WFA_E_INVALID, "id buffer is null");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    WFA_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return WFA_OK;
}

int wfa_rccl_init(wfa_ctx* c, int rank, int n_ranks, const void* id128) {
    if (!c || !id128) return fail(WFA_E_INVALID, "null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(WFA_E_INVALID, "bad rank %d of %d", rank, n_ranks);
    if (c->comm) return fail(WFA_E_STATE, "communicator already initialised");
    WFA_HIP_CHECK(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    WFA_NCCL_CHECK(ncclCommInitRank(&comm, n_ranks, id, rank));
    c->comm = comm;
    c->rank = rank;
    c->n_ranks = n_ranks;
    return WFA_OK;
}

int wfa_rccl_destroy(wfa_ctx* c) {
    if (!c || !c->comm) return WFA_OK;
    (void)hipSetDevice(c->device);
    ncclCommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    return WFA_OK;
}

int wfa_rccl_allgather_counts(wfa_ctx* c, int64_t n_rows, int64_t* counts) {
    if (!c || !c->comm) return fail(WFA_E_STATE, "RCCL communicator not initialised");
    if (!counts || n_rows < 0) return fail(WFA_E_INVALID, "bad counts argument");
    WFA_HIP_CHECK(hipSetDevice(c->device));
    const int n = c->n_ranks;
    DevBuf d_counts;
    int rc = d_counts.ensure((size_t)(n + 1) * sizeof(int64_t));
    if (rc) return rc;
    int64_t* dc = d_counts.as<int64_t>();
    hipError_t e = hipMemcpyAsync(dc + n, &n_rows, sizeof(int64_t), hipMemcpyHostToDevice, c->stream);
    ncclResult_t nr = ncclSuccess;
    if (e == hipSuccess) nr = ncclAllGather(dc + n, dc, 1, ncclInt64, (ncclComm_t)c->comm, c->stream);
    if (e == hipSuccess && nr == ncclSuccess)
        e = hipMemcpyAsync(counts, dc, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && nr == ncclSuccess) e = hipStreamSynchronize(c->stream);
    d_counts.release();
    if (nr != ncclSuccess) return fail(WFA_E_RCCL, "ncclAllGather failed: %s", ncclGetErrorString(nr));
    if (e != hipSuccess) return fail(WFA_E_HIP, "count exchange failed: %s", hipGetErrorString(e));
    return WFA_OK;
}

int wfa_rccl_gather_append(wfa_ctx* c, int on) {
    if (!c) return fail(WFA_E_INVALID, "null context");
    c->gather_append = on != 0;
    if (!on) c->gathered_n = -1;  // the next exchange starts a new table
    return WFA_OK;
}

int wfa_rccl_gather_rows(wfa_ctx* c, const void* rows, int64_t n_rows, int32_t row_bytes, int root,
                         const int64_t* counts, void* out) {
    if (!c || !c->comm) return fail(WFA_E_STATE, "RCCL communicator not initialised");
    if (!counts) return fail(WFA_E_INVALID, "counts is null");
    if (row_bytes <= 0 || n_rows < 0) return fail(WFA_E_INVALID, "bad row geometry");
    if (root < 0 || root >= c->n_ranks) return fail(WFA_E_INVALID, "bad root %d", root);
    if (counts[c->rank] != n_rows) return fail(WFA_E_INVALID, "counts[rank] != n_rows");
    WFA_HIP_CHECK(hipSetDevice(c->device));
    ncclComm_t comm = (ncclComm_t)c->comm;
    const int n = c->n_ranks;

    const uint8_t* d_rows = nullptr;
    DevBuf staged;
    DevBuf& d_all = c->gathered;  // stays resident on the root: the hit-table stages read it (wfa_hit_rows_source(ctx, 2))
    // append mode (wfa_rccl_gather_append): the rows of this exchange go behind the 60-byte rows earlier exchanges left
    const int64_t base = (c->gather_append && row_bytes == 60 && c->gathered_n > 0) ? c->gathered_n : 0;
    c->gathered_n = -1;
    int rc = WFA_OK;
    if (rows == nullptr) {
        if (c->n_hits < 0) return fail(WFA_E_STATE, "no hit pass has been run");
        if (row_bytes != 60 || n_rows != c->n_hits)
            return fail(WFA_E_INVALID, "resident hit rows are %lld x 60 B", (long long)c->n_hits);
        d_rows = c->hit_out.as<uint8_t>();
    } else if (n_rows > 0) {
        if ((rc = staged.ensure((size_t)n_rows * row_bytes))) return rc;
        hipError_t e = hipMemcpyAsync(staged.ptr, rows, (size_t)n_rows * row_bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { staged.release(); return fail(WFA_E_HIP, "H2D failed: %s", hipGetErrorString(e)); }
        d_rows = staged.as<uint8_t>();
    }
    int64_t total = 0;
    for (int r = 0; r < n; ++r) total += counts[r];
    if (c->rank == root) {
        const size_t want = (size_t)(base + total) * row_bytes + 64;
        if (base > 0 && d_all.cap < want) {  // grow without losing the rows already gathered
            DevBuf bigger;
            if ((rc = bigger.ensure(want + want / 2))) { staged.release(); return rc; }
            hipError_t ce = hipMemcpyAsync(bigger.ptr, d_all.ptr, (size_t)base * row_bytes, hipMemcpyDeviceToDevice, c->stream);
            if (ce == hipSuccess) ce = hipStreamSynchronize(c->stream);
            if (ce != hipSuccess) { staged.release(); return fail(WFA_E_HIP, "gather buffer growth failed: %s", hipGetErrorString(ce)); }
            std::swap(d_all.ptr, bigger.ptr);
            std::swap(d_all.cap, bigger.cap);
        } else if ((rc = d_all.ensure(want))) {
            staged.release();
            return rc;
        }
    }
    ncclResult_t nr = ncclGroupStart();
    if (nr == ncclSuccess && n_rows > 0)
        nr = ncclSend(d_rows, (size_t)n_rows * row_bytes, ncclUint8, root, comm, c->stream);
    if (c->rank == root) {
        int64_t at = base;
        for (int r = 0; r < n && nr == ncclSuccess; ++r) {
            if (counts[r] > 0)
                nr = ncclRecv(d_all.as<uint8_t>() + at * row_bytes, (size_t)counts[r] * row_bytes, ncclUint8, r,
                              comm, c->stream);
            at += counts[r];
        }
    }
    ncclResult_t ge = ncclGroupEnd();
    if (nr == ncclSuccess) nr = ge;
    hipError_t e = hipSuccess;
    if (nr == ncclSuccess && c->rank == root && total > 0 && out)  // out == NULL: the rows are only wanted on the device
        e = hipMemcpyAsync(out, d_all.as<uint8_t>() + (size_t)base * row_bytes, (size_t)total * row_bytes,
                           hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = e2;
    staged.release();
    if (nr == ncclSuccess && e == hipSuccess && c->rank == root && row_bytes == 60) c->gathered_n = base + total;
    if (nr != ncclSuccess) return fail(WFA_E_RCCL, "row gather failed: %s", ncclGetErrorString(nr));
    if (e != hipSuccess) return fail(WFA_E_HIP, "row gather failed: %s", hipGetErrorString(e));
    return WFA_OK;
}

}  // extern "C"
