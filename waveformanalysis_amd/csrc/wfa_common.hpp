// Shared host-side definitions of libwfa_hip.so (context, error handling, buffer helpers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "wfa_hip.h"
#include "wfa_kernels.hpp"

namespace wfa {

extern thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...);

#define WFA_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::wfa::fail(WFA_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                          \
    } while (0)

// Growable device buffer.
struct DevBuf {
    void* ptr = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;             // owns device memory
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }                    // whatever wfa_ctx_destroy's list misses is still freed
    int ensure(size_t bytes);  // keeps contents only if no growth is needed
    void release();
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(ptr);
    }
};

struct ProfEntry {
    std::string name;
    double total_ms = 0.0;
    int64_t launches = 0;
};

struct SgPlanDev {
    int W = 0, P = 0, H = 0;
    int n_tables = 0;
    int stride = 0;   // doubles per table
    int int_ok = 0;
    int32_t den = 1, den_edge = 1;
    int64_t guard = 0, guard_edge = 0;
    double rden = 1.0, rden_edge = 1.0;
    DevBuf tab;   // double
    DevBuf itab;  // int32
    DevBuf sym;   // uint8
    std::vector<uint8_t> sym_host;
};

}  // namespace wfa

struct wfa_ctx {
    int device = 0;
    hipStream_t stream = nullptr;

    // resident pool
    wfa::DevBuf pool_u16;
    wfa::DevBuf pool_f32;
    int64_t pool_n = 0;
    bool have_u16 = false, have_f32 = false;

    // records SoA
    int64_t R = 0;
    int32_t max_len = 0;
    bool have_records = false;
    wfa::DevBuf off, len, baseline, pol, thr, ts, dt, board, chan, rid, fixed_bl, bm_off;
    int64_t bitmap_bytes = 0;  // mask bits of all records (k_sg_mask -> k_hit_runs)
    // span mode eligibility (uniform length, contiguous, aligned; see SpanParams)
    bool span_ok = false;
    int32_t span_L = 0;
    int64_t span_off0 = 0;
    int span_positive = 0;
    bool bitmap_clean = false;  // padding bytes of the bitmap are zero

    wfa::SgPlanDev sg;
    bool have_sg = false;
    bool filter_keep = false;  // wfa_filter_keep_output

    // hit scratch
    wfa::DevBuf hit_tmp;        // 60 B rows in chunk order
    int64_t hit_tmp_rows = 0;
    wfa::DevBuf cursor;         // unsigned long long
    wfa::DevBuf rec_tmp_start;  // int64 per record
    wfa::DevBuf rec_nhits;      // int32 per record
    wfa::DevBuf rec_out_start;  // int64 per record
    wfa::DevBuf scan_blocks;    // int64 per scan block
    wfa::DevBuf hit_out;        // final rows
    wfa::DevBuf bitmap;         // 1 bit per sample, per-record regions (bm_off)
    wfa::DevBuf hit_desc;       // int4 (record, start, end, k) per hit
    // streaming pass on uniform records (k_sg_runs32): event buffer, per-span tables, control words
    wfa::DevBuf run_ev, run_span_off, run_span_cnt, run_span_row0, run_scan_blocks, run_ctrl, run_groups, run_lit;
    // host -> device staging: two pinned buffers; a chunk is copied in (a few host threads) while the previous one is on
    // the wire.  Pageable hipMemcpyAsync of a whole pool depends on the driver's own staging (5 GB/s on one box, 0.03
    // GB/s on another)
    void* stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    double last_h2d_GBps = 0.0;  // rate of the last staged upload (wfa_last_h2d_rate)

    // wfa_set_option: choice between code paths that give identical results (tests compare them; a caller never needs to)
    struct {
        bool no_fast = false;      // literal float64 k_hits instead of the integer kernels
        bool no_span = false;      // per-record mask kernel instead of the uniform-record kernels
        bool no_pad = false;       // no padded shadow layout
        bool no_runs32 = false;    // bitmap route (k_sg_mask_span16 + scan + k_hit_runs) instead of k_sg_runs32
        bool no_speculate = false; // exact row launches (host round trip for the hit count)
        bool no_deposit = false;    // streaming kernel: the flush reads the records' last samples from memory again (no LDS deposit)
        int span_records = 0;       // streaming kernel: records per span (measurement; 0 = library's choice)
        bool no_peak_hot = false;   // find_peaks: the plateau machine over every sample (k_find_peaks_staged), no height prefilter
        bool no_peak_slots = false; // find_peaks: count + fill walks instead of one walk into per-record slots
        bool rows_grouped = false;  // hit rows: the 8-lanes-per-hit kernel instead of the flat chunk-per-lane kernel
    } opt;
    wfa::RunsCold* h_cold = nullptr;   // pinned staging (lives behind h_total)
    wfa::RunsCold run_cold_host{};     // what the device copy holds
    bool run_cold_valid = false;
    bool no_runs32 = false;   // a span outgrew the event buffer of its wave: this records upload takes the general route
    int64_t n_hits = -1;
    int64_t last_hits = -1;  // hits of the previous fast-path pass on this context (sizes the speculative tail)

    wfa::DevBuf out_rows;  // per-record feature rows
    wfa::DevBuf out_rows2; // the second table of wfa_features_both (width rows)
    wfa::DevBuf pw_plan;   // numpy pairwise-sum plan of the wave-per-record feature kernels (wfa_features.hip)
    int pw_plan_n = -1;    // reduction length the device copy was built for
    wfa::DevBuf fw_ties;   // width-integral records whose quantile positions are re-walked in numpy's order
    wfa::DevBuf peak_out;  // HIT_DTYPE rows of the last find_peaks pass
    int64_t n_peaks = -1;
    int64_t n_legacy = -1;  // hits of the last wfa_find_hits_count pass
    wfa::DevBuf peak_cand_n, peak_cand_pos, peak_cand_val, peak_cand_state;  // candidate lists of find_peaks
    wfa::DevBuf peak_slot_pos, peak_slot_val;  // kPeakSlots candidates per record, written by the single walk
    wfa::DevBuf peak_cand_rec, peak_accept, peak_ips, peak_row_start;
    wfa::DevBuf wh_pos, wh_row, wh_valid;  // per-hit inputs of k_waveform_width
    // hit-table stages (wfa_hits.hip): scratch slots and the state of the last count pass
    wfa::DevBuf ht[40];
    int ht_src = 1;            // wfa_hit_rows_source: 1 = rows of the last hit pass, 2 = rows of the last gather
    wfa::DevBuf gathered;      // rows the last wfa_rccl_gather_rows left on the root
    int64_t gathered_n = -1;
    bool gather_append = false;  // wfa_rccl_gather_append: exchanges add to the gathered table instead of replacing it
    int64_t ht_n = -1, ht_groups = 0;
    int ht_kind = 0;  // 1 = event grouping, 2 = hit merge, 3 = legacy multi-channel grouping
    int64_t* ht_perm = nullptr;
    // padded device layout for uniform records whose length is not a multiple of 16 samples (the span16 kernels need
    // every lane's 16-sample chunk inside one record): a shadow copy of the u16 pool with the records at stride
    // pad_S = roundup16(L) and the matching offsets column, built on the device the first time a fused pass needs it
    bool pad_ok = false, pad_positive = false, shadow_valid = false;
    int32_t pad_L = 0, pad_S = 0;
    int64_t pad_off0 = 0;
    wfa::DevBuf shadow_pool, shadow_off;
    // CSV decode (wfa_hits.hip): rows / samples of the last wfa_csv_decode_count pass; the samples stay resident
    int64_t csv_rows = -1, csv_samples = -1, csv_bytes = 0;
    int32_t csv_samples_start = 0;
    int csv_delim = ';';
    bool csv_filled = false;
    wfa::DevBuf bw_scratch;  // float64 forward pass of sosfiltfilt, [sample][record-in-batch]

    // profiling: HIP events around every launch on the context's stream.  The pairs are only recorded while the
    // work runs and resolved (hipEventElapsedTime) when the report is read, so timing adds no host round trip
    // between the kernels of a pass.
    bool prof_on = false;
    int prof_level = 0;  // 0 off, 1 every launch, 2 dominant kernels only
    std::vector<wfa::ProfEntry> prof;
    struct PendingEvent {
        hipEvent_t e0, e1;
        std::string name;
    };
    std::vector<PendingEvent> prof_pending;
    std::vector<hipEvent_t> prof_free;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // enqueue-only hit passes (wfa_hits_enqueue / wfa_hits_wait): the row count of the last enqueued pass lands in a
    // pinned host word; the pass's arguments are kept in case it outgrew its speculative row bound and must be redone
    int64_t* h_total = nullptr;
    int64_t run_groups_n = 0;     // group sums that are known to be zero between passes
    bool run_ctrl_clean = false;  // the streaming pass's control words were cleared by the previous pass's last kernel
    bool pending = false;
    struct { int source; bool fused_bl; int32_t bl_start, bl_end, le, re, max_len; int64_t bound; bool runs32; } pend{};

    // rccl (opaque, owned by wfa_rccl.hip)
    void* comm = nullptr;
    int rank = 0, n_ranks = 1;
};

namespace wfa {

// Launch timer: constructed before the launch(es), end(name) after them.
struct LaunchTimer {
    wfa_ctx* c;
    hipEvent_t e0 = nullptr;
    static hipEvent_t take(wfa_ctx* c) {
        if (!c->prof_free.empty()) {
            hipEvent_t e = c->prof_free.back();
            c->prof_free.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    // dominant: the kernel the roofline figure is about -- the only one timed at profile level 2, where the event
    // pairs around the small kernels of a pass would cost as much as the gaps they measure
    explicit LaunchTimer(wfa_ctx* ctx, bool dominant = false) : c(ctx) {
        if (!c->prof_on || (c->prof_level == 2 && !dominant)) return;
        e0 = take(c);
        if (e0) (void)hipEventRecord(e0, c->stream);
    }
    int end(const char* name) {
        if (!c->prof_on || !e0) return WFA_OK;
        hipEvent_t e1 = take(c);
        if (!e1) { c->prof_free.push_back(e0); e0 = nullptr; return WFA_OK; }
        WFA_HIP_CHECK(hipEventRecord(e1, c->stream));
        c->prof_pending.push_back({e0, e1, name});
        e0 = nullptr;
        return WFA_OK;
    }
};

// resolve the recorded pairs into the per-name totals (blocks until the last recorded event has completed)
inline int profile_flush(wfa_ctx* c) {
    for (auto& p : c->prof_pending) {
        float ms = 0.f;
        hipError_t err = hipEventSynchronize(p.e1);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, p.e0, p.e1);
        c->prof_free.push_back(p.e0);
        c->prof_free.push_back(p.e1);
        if (err != hipSuccess) continue;
        bool found = false;
        for (auto& e : c->prof)
            if (e.name == p.name) { e.total_ms += ms; e.launches += 1; found = true; break; }
        if (!found) c->prof.push_back({p.name, (double)ms, 1});
    }
    c->prof_pending.clear();
    return WFA_OK;
}

}  // namespace wfa
