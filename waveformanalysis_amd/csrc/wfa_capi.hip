// C ABI of libwfa_hip.so (see include/wfa_hip.h): context, resident buffers, kernel calls.

#include <cmath>
#include <cstdlib>
#include <new>

#include <algorithm>
#include <chrono>
#include <thread>

#include "wfa_common.hpp"
#include "wfa_host.hpp"
#include "wfa_kernels.hpp"

namespace wfa {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int DevBuf::ensure(size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (bytes <= cap) return WFA_OK;
    if (ptr) {
        (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    // 256 B of slack so 16-byte vector loads at the tail stay inside the allocation
    hipError_t e = hipMalloc(&ptr, bytes + 256);
    if (e != hipSuccess) {
        ptr = nullptr;
        return fail(WFA_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    cap = bytes;
    return WFA_OK;
}

void DevBuf::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

static int use_device(wfa_ctx* c) {
    if (!c) return fail(WFA_E_INVALID, "null context");
    WFA_HIP_CHECK(hipSetDevice(c->device));
    return WFA_OK;
}

constexpr size_t kStageBytes = 32u << 20;   // per staging buffer
constexpr size_t kStageMin = 4u << 20;      // smaller copies go straight through hipMemcpyAsync

// host -> device through the context's pinned double buffer (see wfa_ctx::stage; the ring itself: wfa_host.hpp); returns
// when the last chunk has landed
static int h2d_staged(wfa_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c->stage[0]) {
        for (int b = 0; b < 2; ++b) {
            WFA_HIP_CHECK(hipHostMalloc(&c->stage[b], kStageBytes, hipHostMallocDefault));
            WFA_HIP_CHECK(hipEventCreateWithFlags(&c->stage_ev[b], hipEventDisableTiming));
        }
        c->stage_bytes = kStageBytes;
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t err = hipSuccess;
    const int rc = host::staged_copy(
        dst, src, bytes, c->stage, kStageBytes,
        [&](void* d, const void* staged, size_t n, int b) {
            err = hipMemcpyAsync(d, staged, n, hipMemcpyHostToDevice, c->stream);
            if (err == hipSuccess) err = hipEventRecord(c->stage_ev[b], c->stream);
            return err == hipSuccess ? 0 : 1;
        },
        [&](int b) { return (err = hipEventSynchronize(c->stage_ev[b])) == hipSuccess ? 0 : 1; });
    if (rc) return fail(WFA_E_HIP, "staged upload failed: %s", hipGetErrorString(err));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));  // the staging buffers are free again for the next call
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (s > 0) c->last_h2d_GBps = (double)bytes / s / 1e9;
    return WFA_OK;
}

static int h2d(wfa_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    int rc = b.ensure(bytes);
    if (rc) return rc;
    if (bytes >= kStageMin) return h2d_staged(c, b.ptr, src, bytes);
    if (bytes) WFA_HIP_CHECK(hipMemcpyAsync(b.ptr, src, bytes, hipMemcpyHostToDevice, c->stream));
    return WFA_OK;
}

static PoolView pool_view(wfa_ctx* c) {
    PoolView p;
    p.u16 = c->have_u16 ? c->pool_u16.as<uint16_t>() : nullptr;
    p.f32 = c->have_f32 ? c->pool_f32.as<float>() : nullptr;
    p.n = c->pool_n;
    return p;
}

static RecView rec_view(wfa_ctx* c) {
    RecView r;
    r.R = c->R;
    r.off = c->off.as<int64_t>();
    r.len = c->len.as<int32_t>();
    r.baseline = c->baseline.as<double>();
    r.baseline_rw = c->baseline.as<double>();
    r.pol = c->pol.as<int8_t>();
    r.thr = c->thr.as<double>();
    r.ts = c->ts.as<int64_t>();
    r.dt = c->dt.as<int32_t>();
    r.board = c->board.as<int16_t>();
    r.chan = c->chan.as<int16_t>();
    r.rid = c->rid.as<int64_t>();
    r.bm_off = c->bm_off.as<int64_t>();
    return r;
}

static SgParams sg_params(wfa_ctx* c) {
    SgParams s{};
    if (!c->have_sg) return s;
    s.W = c->sg.W; s.P = c->sg.P; s.H = c->sg.H; s.stride = c->sg.stride;
    s.tab = c->sg.tab.as<double>();
    s.sym = c->sg.sym.as<uint8_t>();
    s.int_ok = c->sg.int_ok;
    s.itab = c->sg.itab.as<int32_t>();
    s.den = c->sg.den; s.den_edge = c->sg.den_edge;
    s.guard = c->sg.guard; s.guard_edge = c->sg.guard_edge;
    s.rden = c->sg.rden; s.rden_edge = c->sg.rden_edge;
    // |y| < 2^17 for uint16 samples (sum|c| < 2): float32 ulp <= 2^-7
    s.margin = (int32_t)((c->sg.den + 127) / 128 + 2);
    s.margin_edge = (int32_t)((c->sg.den_edge + 127) / 128 + 2);
    return s;
}

static int need_source(wfa_ctx* c, int source) {
    if (!c->have_records) return fail(WFA_E_STATE, "records not uploaded");
    switch (source) {
        case WFA_SRC_RAW:
            if (!c->have_u16) return fail(WFA_E_STATE, "wave_pool (uint16) not uploaded");
            return WFA_OK;
        case WFA_SRC_F32:
            if (!c->have_f32) return fail(WFA_E_STATE, "wave_pool_filtered (float32) not resident");
            return WFA_OK;
        case WFA_SRC_SG_FUSED:
            if (!c->have_u16) return fail(WFA_E_STATE, "wave_pool (uint16) not uploaded");
            if (!c->have_sg) return fail(WFA_E_STATE, "Savitzky-Golay plan not set");
            return WFA_OK;
        default:
            return fail(WFA_E_INVALID, "unknown wave source %d", source);
    }
}

// padded shadow of the u16 pool (see wfa_ctx::pad_ok): built once per upload, on the device
static int ensure_shadow(wfa_ctx* c) {
    if (c->shadow_valid) return WFA_OK;
    int rc;
    const size_t shadow_samples = (size_t)c->R * c->pad_S;
    if ((rc = c->shadow_pool.ensure(shadow_samples * sizeof(uint16_t) + 256))) return rc;
    if ((rc = c->shadow_off.ensure((size_t)c->R * sizeof(int64_t)))) return rc;
    LaunchTimer t(c);
    WFA_HIP_CHECK(hipMemsetAsync(c->shadow_pool.as<uint8_t>() + shadow_samples * sizeof(uint16_t), 0, 256, c->stream));
    WFA_HIP_CHECK(launch_pad_rows(c->stream, c->pool_u16.as<uint16_t>(), c->pad_off0, c->pad_L, c->pad_S, c->R,
                                  c->shadow_pool.as<uint16_t>(), c->shadow_off.as<int64_t>()));
    if ((rc = t.end("k_pad_rows (once per upload)"))) return rc;
    c->shadow_valid = true;
    return WFA_OK;
}


// Uniform records (span mode), Savitzky-Golay source: streaming kernel with ordered run events (wfa_stream.hip).
//   k_sg_runs32 -> scan of the per-span hit counts -> k_runs_to_desc -> row kernels
// *done = false: the layout / options are outside what that kernel covers, or a span outgrew its event buffer --
// the caller takes the general route (per-record mask kernel + bitmap).
static int run_hits_runs32(wfa_ctx* c, bool fused_bl, int32_t bl_start, int32_t bl_end, int32_t le, int32_t re,
                           int32_t max_len, int64_t* n_hits, bool enqueue_only, bool* done) {
    *done = false;
    if (c->no_runs32 || c->opt.no_runs32) return WFA_OK;
    const SgParams sp0 = sg_params(c);
    SpanParams sp{};
    bool padded = false;
    if (c->span_ok && c->span_L % 32 == 0) {
        sp.off0 = c->span_off0; sp.L = c->span_L; sp.S = c->span_L; sp.positive = c->span_positive;
    } else if (c->pad_ok && c->pad_S % 32 == 0 && !c->opt.no_pad) {
        padded = true;
        sp.off0 = 0; sp.L = c->pad_L; sp.S = c->pad_S; sp.positive = c->pad_positive;
    } else {
        return WFA_OK;
    }
    if (!sg_runs32_supported(sp0, sp.L, sp.S, bl_start, bl_end, fused_bl)) return WFA_OK;
    int rc;
    const int64_t R = c->R;
    int32_t g_wstride = 0, g_nseg = 0, g_segw = 0;
    sg_runs32_geometry(sp.S, &sp.rs, &g_wstride, &g_nseg, &g_segw);
    if (c->opt.span_records > 0 && c->opt.span_records < sp.rs) sp.rs = c->opt.span_records;
    sp.n_spans = (R + sp.rs - 1) / sp.rs;
    const int64_t ns = sp.n_spans;
    const int64_t nb = scan_blocks_for(ns);
    if ((rc = c->run_span_off.ensure((size_t)ns * sizeof(int64_t)))) return rc;
    if ((rc = c->run_span_cnt.ensure((size_t)ns * sizeof(int32_t)))) return rc;
    if ((rc = c->run_span_row0.ensure((size_t)ns * sizeof(int64_t)))) return rc;
    if ((rc = c->run_scan_blocks.ensure((size_t)(nb + 1) * sizeof(int64_t)))) return rc;
    const int64_t n_groups = (ns + 63) / 64;
    if (c->run_groups.cap < (size_t)n_groups * sizeof(unsigned long long)) c->run_groups_n = 0;  // (a new allocation is not zero)
    if ((rc = c->run_groups.ensure((size_t)n_groups * sizeof(unsigned long long)))) return rc;
    if (n_groups > c->run_groups_n) c->run_ctrl_clean = false;  // sums beyond what has ever been cleared
    if (c->run_ctrl.cap < 256 + sizeof(RunsCold)) { c->run_cold_valid = false; c->run_ctrl_clean = false; }
    if ((rc = c->run_ctrl.ensure(256 + sizeof(RunsCold)))) return rc;

    PoolView pvf = pool_view(c);
    RecView rvf = rec_view(c);
    if (padded) {
        if ((rc = ensure_shadow(c))) return rc;
        pvf.u16 = c->shadow_pool.as<uint16_t>();
        pvf.n = c->R * (int64_t)c->pad_S;
        rvf.off = c->shadow_off.as<int64_t>();
    }
    RowParams rp{le, re, max_len, sp0.W / 2};
    rp.uni_L = sp.L; rp.uni_S = sp.S == sp.L ? 0 : sp.S; rp.uni_positive = sp.positive ? 1 : 0; rp.uni_off0 = sp.off0;
    int64_t* d_total = c->run_scan_blocks.as<int64_t>() + nb;
    auto* ctrl = c->run_ctrl.as<unsigned long long>();  // [0] event cursor, [1] flags, [2] hits listed for the literal kernel
    constexpr int kLitCap = 65536;
    if ((rc = c->run_lit.ensure((size_t)kLitCap * sizeof(int32_t)))) return rc;
    if (!c->opt.rows_grouped) {  // (the 8-lanes-per-hit kernel only flags)
        rp.lit_cnt = reinterpret_cast<uint32_t*>(ctrl + 2);
        rp.lit_list = c->run_lit.as<int32_t>();
        rp.lit_cap = kLitCap;
    }

    for (int attempt = 0; attempt < 3; ++attempt) {
        // speculative tail (see run_hits): row buffers sized from the previous pass on this context
        const int64_t held = (int64_t)std::min(c->hit_out.cap / 60, c->hit_desc.cap / sizeof(int4));
        const int64_t bound = c->last_hits + c->last_hits / 8 + 4096;
        const bool spec = c->last_hits >= 0 && held >= bound && !c->opt.no_speculate && attempt == 0;
        // every span owns a fixed slot of the event buffer (8 KiB: 160 MB per 10^9 samples, only the lines that hold
        // events are ever touched)
        const int64_t ev_want = ns * sg_runs32_event_slot();
        if ((int64_t)(c->run_ev.cap / sizeof(uint32_t)) < ev_want)
            if ((rc = c->run_ev.ensure((size_t)ev_want * sizeof(uint32_t)))) return rc;
        RunsParams rn{};
        rn.ev = c->run_ev.as<uint32_t>();
        rn.ev_cap = (int64_t)(c->run_ev.cap / sizeof(uint32_t));
        rn.cursor = ctrl;
        rn.flags = reinterpret_cast<int32_t*>(ctrl + 1);
        rn.span_off = c->run_span_off.as<int64_t>();
        rn.span_cnt = c->run_span_cnt.as<int32_t>();
        rn.group_sum = c->run_groups.as<unsigned long long>();
        rn.lit_cnt = c->opt.rows_grouped ? nullptr : reinterpret_cast<uint32_t*>(ctrl + 2);
        // the control words are cleared by the last kernel of a queued pass (RowParams::pass_ctrl); a memset only when the
        // pass before did not end that way
        if (!c->run_ctrl_clean) {
            WFA_HIP_CHECK(hipMemsetAsync(ctrl, 0, 16, c->stream));
            WFA_HIP_CHECK(hipMemsetAsync(c->run_groups.ptr, 0, (size_t)n_groups * sizeof(unsigned long long), c->stream));
            if (n_groups > c->run_groups_n) c->run_groups_n = n_groups;
        }
        c->run_ctrl_clean = false;
        RunsArgs ra{};
        ra.pool = pvf.u16; ra.thr = rvf.thr; ra.baseline = rvf.baseline_rw; ra.R = R;
        ra.itab = sp0.itab; ra.den = sp0.den; ra.margin = sp0.margin;
        ra.den_edge = sp0.den_edge; ra.margin_edge = sp0.margin_edge;
        // sg_plan.py: guard = 8 eps den^2 2^24 + 1 with eps = bound on |scipy's float64 chain - exact rational|; here in
        // numerator units with a factor 4 of head room (and never below 1e-6)
        ra.delta = std::max(4.0 * (double)sp0.guard / (8.0 * (double)sp0.den * 16777216.0), 1e-6);
        ra.W = sp0.W; ra.L = sp.L; ra.S = sp.S; ra.positive = sp.positive; ra.rs = sp.rs;
        ra.wstride = g_wstride; ra.nseg = g_nseg; ra.segw = g_segw;
        ra.dep = (!c->opt.no_deposit && sg_runs32_deposit(sp.L, sp.S, sp0.W, sp.rs, g_wstride)) ? 1 : 0;
        ra.off0 = sp.off0; ra.n_spans = ns;
        ra.ev = rn.ev; ra.ev_cap = rn.ev_cap; ra.cursor = rn.cursor; ra.span_off = rn.span_off; ra.span_cnt = rn.span_cnt;
        ra.flags = rn.flags;
        ra.group_sum = rn.group_sum;
        ra.cold = reinterpret_cast<const RunsCold*>(ctrl + 32);  // 256 bytes behind the atomically updated words
        {
            // what the float64 reference paths read: a device copy next to the control words, refreshed when it changes
            RunsCold cold{pvf, sp0};
            if (!c->run_cold_valid || memcmp(&cold, &c->run_cold_host, sizeof(cold)) != 0) {
                memcpy(c->h_cold, &cold, sizeof(cold));
                WFA_HIP_CHECK(hipMemcpyAsync(ctrl + 32, c->h_cold, sizeof(cold), hipMemcpyHostToDevice, c->stream));
                WFA_HIP_CHECK(hipStreamSynchronize(c->stream));  // h_cold may be rewritten by the next pass
                c->run_cold_host = cold;
                c->run_cold_valid = true;
            }
        }
        {
            LaunchTimer t(c, true);
            WFA_HIP_CHECK(launch_sg_runs32(c->stream, fused_bl, ra));
            if ((rc = t.end(fused_bl ? "k_sg_runs32<baseline>" : "k_sg_runs32"))) return rc;
        }
        auto rows = [&](int64_t n_rows) -> int {
            {
                LaunchTimer t(c);
                WFA_HIP_CHECK(launch_runs_to_desc(c->stream, rn, ns, sp.rs, d_total, rp.cap, c->hit_desc.as<int4>()));
                if (int r2 = t.end("k_runs_to_desc")) return r2;
            }
            {
                LaunchTimer t(c);
                WFA_HIP_CHECK(launch_hit_rows_fast(c->stream, pvf, rvf, sp0, rp, c->hit_desc.as<int4>(), n_rows,
                                                   c->hit_out.as<uint8_t>(), c->opt.rows_grouped));
                if (int r2 = t.end(c->opt.rows_grouped ? "k_hit_rows_grp" : "k_hit_rows_flat")) return r2;
            }
            {
                LaunchTimer t(c);
                WFA_HIP_CHECK(launch_hit_rows_literal(c->stream, WFA_SRC_SG_FUSED, pvf, rvf, sp0, rp,
                                                      c->hit_desc.as<int4>(), n_rows, true, c->hit_out.as<uint8_t>()));
                if (int r2 = t.end("k_hit_rows_literal")) return r2;
            }
            return WFA_OK;
        };
        int64_t total = 0;
        unsigned long long ctl[2] = {0, 0};
        if (spec && enqueue_only) {
            // total, cursor and flags are written to the pinned words by the pass's last kernel, which also clears the
            // control words for the next pass; nobody waits here
            rp.cap = bound;
            rp.n_dev = d_total;
            rp.pass_report = c->h_total;
            rp.pass_ctrl = ctrl;
            rp.pass_groups = rn.group_sum;
            rp.pass_n_groups = n_groups;
            if ((rc = rows(bound))) return rc;
            c->run_ctrl_clean = true;
            {
                c->pending = true;
                c->pend = {WFA_SRC_SG_FUSED, fused_bl, bl_start, bl_end, le, re, max_len, bound, true};
                c->n_hits = -1;
                *done = true;
                return WFA_OK;
            }
        }
        if (spec) {
            rp.cap = bound;
            rp.n_dev = d_total;
            if ((rc = rows(bound))) return rc;
        } else {
            // no speculative row launch (first pass on this context, buffers too small): the host needs the row count
            // before it can size and launch the rows -- the one case that still scans the span counts
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_scan(c->stream, rn.span_cnt, ns, c->run_scan_blocks.as<int64_t>(),
                                      c->run_span_row0.as<int64_t>()));
            if ((rc = t.end("k_scan(span hit counts)"))) return rc;
        }
        WFA_HIP_CHECK(hipMemcpyAsync(&total, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipMemcpyAsync(ctl, ctrl, 16, hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        const int flags = (int)(ctl[1] & 0xffffffffull);
        if (flags & 1) {  // a span holds more events than a wave buffers: general route for this upload
            c->no_runs32 = true;
            return WFA_OK;
        }
        if (flags & 2) {  // a span's patched events outgrew its slot (cannot happen for W <= 11): general route
            c->no_runs32 = true;
            return WFA_OK;
        }
        c->last_hits = total;
        if (spec && total <= bound) {
            c->n_hits = total;
            *n_hits = total;
            *done = true;
            return WFA_OK;
        }
        // first pass on this context, or more hits than the guess: exact sizes (with head room for the next pass)
        rp.cap = 0;
        rp.n_dev = nullptr;
        const int64_t want = total + total / 8 + 4096;
        if ((rc = c->hit_out.ensure((size_t)want * 60))) return rc;
        if ((rc = c->hit_desc.ensure((size_t)want * sizeof(int4)))) return rc;
        if ((rc = rows(total))) return rc;
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->n_hits = total;
        *n_hits = total;
        *done = true;
        return WFA_OK;
    }
    return fail(WFA_E_NOMEM, "event buffer of the streaming hit pass did not converge");
}

// enqueue_only: when the pass can take the speculative row launch, queue it and return without waiting for the row
// count (n_hits may be null; wfa_hits_wait delivers it); otherwise the pass runs to completion as usual.
static int run_hits(wfa_ctx* c, int source, bool fused_bl, int32_t bl_start, int32_t bl_end,
                    int32_t le, int32_t re, int32_t max_len, int64_t* n_hits, bool enqueue_only = false) {
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = need_source(c, source))) return rc;
    if (!n_hits && !enqueue_only) return fail(WFA_E_INVALID, "n_hits is null");
    int64_t n_hits_local = 0;
    if (!n_hits) n_hits = &n_hits_local;
    c->pending = false;
    if (le < 0) le = 0;  // hit_finder.py:124-125
    if (re < 0) re = 0;
    if (max_len <= 0) max_len = c->max_len;
    if (max_len < c->max_len)
        return fail(WFA_E_INVALID, "max_len (%d) < longest uploaded record (%d)", max_len, c->max_len);
    if (fused_bl && bl_start < 0) return fail(WFA_E_INVALID, "baseline window start must be >= 0");
    c->n_hits = -1;
    if (c->R == 0) {
        c->n_hits = 0;
        *n_hits = 0;
        return WFA_OK;
    }

    const int64_t R = c->R;
    if ((rc = c->rec_tmp_start.ensure(R * sizeof(int64_t)))) return rc;
    if ((rc = c->rec_nhits.ensure(R * sizeof(int32_t)))) return rc;
    if ((rc = c->rec_out_start.ensure(R * sizeof(int64_t)))) return rc;
    const int64_t nb = scan_blocks_for(R);
    if ((rc = c->scan_blocks.ensure((nb + 1) * sizeof(int64_t)))) return rc;
    if ((rc = c->cursor.ensure(sizeof(unsigned long long)))) return rc;

    const PoolView pv0 = pool_view(c);
    const RecView rv0 = rec_view(c);
    const SgParams sp0 = sg_params(c);
    if (source == WFA_SRC_SG_FUSED && sg_mask_supported(sp0) && !c->opt.no_fast) {
        {
            bool done = false;
            if ((rc = run_hits_runs32(c, fused_bl, bl_start, bl_end, le, re, max_len, n_hits, enqueue_only, &done))) return rc;
            if (done) return WFA_OK;
        }
        // A: mask + run counts  ->  scan  ->  B1: run descriptors  ->  B2: rows (final order, no gather)
        if (c->bitmap.cap < (size_t)c->bitmap_bytes) c->bitmap_clean = false;
        if ((rc = c->bitmap.ensure((size_t)c->bitmap_bytes))) return rc;
        if (!c->bitmap_clean) {  // bytes between a record's last mask byte and its next word stay zero
            WFA_HIP_CHECK(hipMemsetAsync(c->bitmap.ptr, 0, (size_t)c->bitmap_bytes, c->stream));
            c->bitmap_clean = true;
        }
        MaskParams mp{};
        mp.bl_start = bl_start; mp.bl_end = fused_bl ? bl_end : bl_start;
        mp.bitmap = c->bitmap.as<uint8_t>();
        mp.rec_nhits = c->rec_nhits.as<int32_t>();
        // views the kernels of this pass read: the uploaded layout, or the padded shadow of it
        PoolView pvf = pv0;
        RecView rvf = rv0;
        const bool padded = c->pad_ok && sg_mask_span16_padded_supported(sp0, c->pad_L) && !c->opt.no_pad &&
                            !c->opt.no_span;
        if (padded) {
            if ((rc = ensure_shadow(c))) return rc;
            pvf.u16 = c->shadow_pool.as<uint16_t>();
            pvf.n = c->R * (int64_t)c->pad_S;
            rvf.off = c->shadow_off.as<int64_t>();
        }
        if (padded || (c->span_ok && c->span_L >= sp0.W && !c->opt.no_span)) {
            SpanParams sp{};
            sp.off0 = c->span_off0; sp.L = c->span_L; sp.positive = c->span_positive;
            if (padded) { sp.off0 = 0; sp.L = c->pad_L; sp.S = c->pad_S; sp.positive = c->pad_positive; }
            const int32_t span_L = sp.L;
            sp.rs = 64;
            sp.n_spans = (R + sp.rs - 1) / sp.rs;
            sp.bm_off0 = 0;
            sp.bm_stride = ((int64_t)span_L + 7 + 63) / 64 * 8 + 8;
            LaunchTimer t(c, true);
            if (padded || (sg_mask_span16_supported(sp0, span_L))) {
                WFA_HIP_CHECK(launch_sg_mask_span16(c->stream, fused_bl, pvf, rvf, sp0, mp, sp));
                if ((rc = t.end(fused_bl ? "k_sg_mask_span16<baseline>" : "k_sg_mask_span16"))) return rc;
            } else {
                WFA_HIP_CHECK(launch_sg_mask(c->stream, fused_bl, c->max_len, pvf, rvf, sp0, mp));
                if ((rc = t.end(fused_bl ? "k_sg_mask<baseline>" : "k_sg_mask"))) return rc;
            }
        } else {
            LaunchTimer t(c, true);
            WFA_HIP_CHECK(launch_sg_mask(c->stream, fused_bl, c->max_len, pvf, rvf, sp0, mp));
            if ((rc = t.end(fused_bl ? "k_sg_mask<baseline>" : "k_sg_mask"))) return rc;
        }
        {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_scan(c->stream, c->rec_nhits.as<int32_t>(), R, c->scan_blocks.as<int64_t>(),
                                      c->rec_out_start.as<int64_t>()));
            if ((rc = t.end("k_scan(hit counts)"))) return rc;
        }
        int64_t total = 0;
        const int64_t* d_total = c->scan_blocks.as<int64_t>() + nb;
        RowParams rp{le, re, max_len, sp0.W / 2};
        if (padded) { rp.uni_L = c->pad_L; rp.uni_S = c->pad_S; rp.uni_positive = c->pad_positive ? 1 : 0; rp.uni_off0 = 0; }
        else if (c->span_ok) { rp.uni_L = c->span_L; rp.uni_positive = c->span_positive ? 1 : 0; rp.uni_off0 = c->span_off0; }
        {
            // mask bytes of one block's records (+16 for the aligned start), rounded up to 1 KiB
            const int64_t per_rec = ((int64_t)c->max_len + 7 + 63) / 64 * 8 + 8;
            const int64_t need = (per_rec * hit_runs_block() + 16 + 1023) / 1024 * 1024;
            rp.stage_bytes = need <= 48 * 1024 ? (int32_t)need : 48 * 1024;
        }
        // Speculative tail: a pass usually finds about as many hits as the previous one on this context.  When the
        // row buffers already hold that many (+12 %), the row kernels are launched for that bound straight away and
        // take the real count from the device; the host reads it once, after everything is queued.  A pass that
        // finds more is redone the exact way below.
        const int64_t held = (int64_t)std::min(c->hit_out.cap / 60, c->hit_desc.cap / sizeof(int4));
        const int64_t bound = c->last_hits + c->last_hits / 8 + 4096;
        if (c->last_hits >= 0 && held >= bound && !c->opt.no_speculate) {
            rp.cap = bound;
            rp.n_dev = d_total;
            {
                LaunchTimer t(c);
                WFA_HIP_CHECK(launch_hit_runs(c->stream, rvf, c->bitmap.as<uint8_t>(), c->rec_nhits.as<int32_t>(),
                                              c->rec_out_start.as<int64_t>(), c->hit_desc.as<int4>(), rp));
                if ((rc = t.end("k_hit_runs"))) return rc;
            }
            {
                LaunchTimer t(c);
                WFA_HIP_CHECK(launch_hit_rows_fast(c->stream, pvf, rvf, sp0, rp, c->hit_desc.as<int4>(), bound,
                                                   c->hit_out.as<uint8_t>(), c->opt.rows_grouped));
                if ((rc = t.end(c->opt.rows_grouped ? "k_hit_rows_grp" : "k_hit_rows_flat"))) return rc;
            }
            {
                LaunchTimer t(c);
                WFA_HIP_CHECK(launch_hit_rows_literal(c->stream, WFA_SRC_SG_FUSED, pvf, rvf, sp0, rp,
                                                      c->hit_desc.as<int4>(), bound, true, c->hit_out.as<uint8_t>()));
                if ((rc = t.end("k_hit_rows_literal"))) return rc;
            }
            if (enqueue_only) {  // the count goes to the pinned word; nobody waits here
                WFA_HIP_CHECK(hipMemcpyAsync(c->h_total, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
                c->pending = true;
                c->pend = {source, fused_bl, bl_start, bl_end, le, re, max_len, bound, false};
                c->n_hits = -1;
                return WFA_OK;
            }
            WFA_HIP_CHECK(hipMemcpyAsync(&total, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
            WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (total <= bound) {
                c->last_hits = total;
                c->n_hits = total;
                *n_hits = total;
                return WFA_OK;
            }
            rp.cap = 0;
            rp.n_dev = nullptr;
        }
        WFA_HIP_CHECK(hipMemcpyAsync(&total, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->last_hits = total;
        // first pass on this context (or a pass that outgrew the guess): size the row buffers with head room so
        // that the next pass can take the speculative tail
        const int64_t want = total + total / 8 + 4096;
        if ((rc = c->hit_out.ensure((size_t)want * 60))) return rc;
        if ((rc = c->hit_desc.ensure((size_t)want * sizeof(int4)))) return rc;
        {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_hit_runs(c->stream, rvf, c->bitmap.as<uint8_t>(), c->rec_nhits.as<int32_t>(),
                                          c->rec_out_start.as<int64_t>(), c->hit_desc.as<int4>(), rp));
            if ((rc = t.end("k_hit_runs"))) return rc;
        }
        {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_hit_rows_fast(c->stream, pvf, rvf, sp0, rp, c->hit_desc.as<int4>(), total,
                                               c->hit_out.as<uint8_t>(), c->opt.rows_grouped));
            if ((rc = t.end(c->opt.rows_grouped ? "k_hit_rows_grp" : "k_hit_rows_flat"))) return rc;
        }
        {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_hit_rows_literal(c->stream, WFA_SRC_SG_FUSED, pvf, rvf, sp0, rp,
                                                  c->hit_desc.as<int4>(), total, true, c->hit_out.as<uint8_t>()));
            if ((rc = t.end("k_hit_rows_literal"))) return rc;
        }
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->n_hits = total;
        *n_hits = total;
        return WFA_OK;
    }

    HitParams hp{};
    hp.le = le; hp.re = re; hp.max_len = max_len;
    hp.bl_start = bl_start; hp.bl_end = bl_end;
    hp.bm_words = (c->max_len + 7 + 63) / 64 + 9;  // bits are indexed from the 16-byte aligned base
    hp.chunk_rows = 256;
    hp.use_fast = c->opt.no_fast ? 0 : 1;
    const int64_t waves = hits_waves(R);
    int64_t want_rows = waves * hp.chunk_rows + c->pool_n / 256 + 4096;
    if (c->hit_tmp_rows < want_rows) {
        if ((rc = c->hit_tmp.ensure((size_t)want_rows * 60))) return rc;
        c->hit_tmp_rows = want_rows;
    }

    const PoolView pv = pool_view(c);
    const RecView rv = rec_view(c);
    const SgParams sp = sg_params(c);
    for (int attempt = 0; attempt < 2; ++attempt) {
        hp.tmp = c->hit_tmp.as<uint8_t>();
        hp.tmp_rows = c->hit_tmp_rows;
        hp.cursor = c->cursor.as<unsigned long long>();
        hp.rec_tmp_start = c->rec_tmp_start.as<int64_t>();
        hp.rec_nhits = c->rec_nhits.as<int32_t>();
        WFA_HIP_CHECK(hipMemsetAsync(c->cursor.ptr, 0, sizeof(unsigned long long), c->stream));
        {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_hits(c->stream, source, fused_bl, pv, rv, sp, hp));
            const char* name = source == WFA_SRC_SG_FUSED
                                   ? (fused_bl ? "k_hits<sg_fused,baseline>" : "k_hits<sg_fused>")
                                   : (source == WFA_SRC_F32 ? "k_hits<f32>" : (fused_bl ? "k_hits<raw,baseline>" : "k_hits<raw>"));
            if ((rc = t.end(name))) return rc;
        }
        unsigned long long used = 0;
        WFA_HIP_CHECK(hipMemcpyAsync(&used, c->cursor.ptr, sizeof(used), hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        if ((int64_t)used <= c->hit_tmp_rows) break;
        if (attempt == 1) return fail(WFA_E_NOMEM, "hit scratch overflow after regrow");
        // chunk placement is deterministic for a fixed grid: `used` rows are enough
        const int64_t rows = (int64_t)used + 1024;
        if ((rc = c->hit_tmp.ensure((size_t)rows * 60))) return rc;
        c->hit_tmp_rows = rows;
    }

    {
        LaunchTimer t(c);
        WFA_HIP_CHECK(launch_scan(c->stream, c->rec_nhits.as<int32_t>(), R, c->scan_blocks.as<int64_t>(),
                                  c->rec_out_start.as<int64_t>()));
        if ((rc = t.end("k_scan(hit counts)"))) return rc;
    }
    int64_t total = 0;
    WFA_HIP_CHECK(hipMemcpyAsync(&total, c->scan_blocks.as<int64_t>() + nb, sizeof(int64_t),
                                 hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    if ((rc = c->hit_out.ensure((size_t)total * 60))) return rc;
    {
        LaunchTimer t(c);
        WFA_HIP_CHECK(launch_hits_gather(c->stream, c->hit_tmp.as<uint8_t>(), c->rec_tmp_start.as<int64_t>(),
                                         c->rec_nhits.as<int32_t>(), c->rec_out_start.as<int64_t>(), R,
                                         c->hit_out.as<uint8_t>()));
        if ((rc = t.end("k_hits_gather"))) return rc;
    }
    c->n_hits = total;
    *n_hits = total;
    return WFA_OK;
}

}  // namespace wfa

using namespace wfa;

// ---- records as packed rows, unpacked on the device --------------------------------------------------------------------
// (reference row layout: processing/dtypes.py:80-100, 102 bytes, numpy packs without alignment.)  The column route above
// makes the caller extract ten columns on the host -- for 1.25e6 records that is 0.6 s of numpy, 0.47 s of it comparing
// the 32-byte `polarity` strings -- before 1 ms of GPU work.  Here the rows go through the pinned staging ring as they are
// (128 MB: 3 ms) and one kernel writes the columns, validates them and finds what the host loop above finds.
struct PackedFields {  // byte offsets inside a row; -1 = field absent (default used)
    int32_t off, len, baseline, polarity, ts, dt, board, chan, rid;
    int32_t polarity_chars;  // width of the UCS-4 polarity field in characters (>= 8)
    int32_t row_bytes;
};
struct UnpackResult {  // one per upload, device -> host
    unsigned long long first_bad[4];  // first record with: negative offset, negative length, out of pool, too long
    int max_len;
    int nonuniform;       // some record breaks (len == len0, off == off0 + r len0, polarity class == class 0)
    int rid_not_increasing;
    int pad;
};

template <typename T>
__device__ __forceinline__ T load_unaligned(const uint8_t* p) {
    T v;
    uint8_t b[sizeof(T)];
#pragma unroll
    for (size_t i = 0; i < sizeof(T); ++i) b[i] = p[i];
    __builtin_memcpy(&v, b, sizeof(T));
    return v;
}

__device__ __forceinline__ bool ucs4_equals(const uint8_t* p, int chars, const char* word) {  // `word` has 8 letters
    bool eq = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) eq = eq && load_unaligned<uint32_t>(p + 4 * i) == (uint32_t)(unsigned char)word[i];
    if (chars > 8) eq = eq && load_unaligned<uint32_t>(p + 32) == 0u;
    return eq;
}

__global__ __launch_bounds__(256) void k_unpack_records(const uint8_t* __restrict__ rows, int64_t R, PackedFields f,
                                                         int64_t pool_n, double thr_scalar, const double* __restrict__ thr_in,
                                                         const int8_t* __restrict__ pol_in, RecView out_cols,
                                                         int8_t* __restrict__ out_pol, double* __restrict__ out_thr,
                                                         int64_t* __restrict__ out_off, int32_t* __restrict__ out_len,
                                                         int64_t* __restrict__ out_ts, int32_t* __restrict__ out_dt,
                                                         int16_t* __restrict__ out_board, int16_t* __restrict__ out_chan,
                                                         int64_t* __restrict__ out_rid, int32_t* __restrict__ bm_bytes,
                                                         UnpackResult* __restrict__ res) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const uint8_t* row = rows + r * f.row_bytes;
    const int64_t off = load_unaligned<int64_t>(row + f.off);
    const int32_t len = load_unaligned<int32_t>(row + f.len);
    auto pol_of = [&](const uint8_t* rw) -> int8_t {
        if (f.polarity < 0) return WFA_POL_UNKNOWN;
        if (ucs4_equals(rw + f.polarity, f.polarity_chars, "negative")) return WFA_POL_NEGATIVE;
        if (ucs4_equals(rw + f.polarity, f.polarity_chars, "positive")) return WFA_POL_POSITIVE;
        return WFA_POL_UNKNOWN;
    };
    const int8_t pol = pol_in ? pol_in[r] : pol_of(row);
    const int64_t rid = load_unaligned<int64_t>(row + f.rid);
    out_off[r] = off;
    out_len[r] = len;
    out_cols.baseline_rw[r] = load_unaligned<double>(row + f.baseline);
    out_pol[r] = pol;
    out_thr[r] = thr_in ? thr_in[r] : thr_scalar;
    out_ts[r] = load_unaligned<int64_t>(row + f.ts);
    out_dt[r] = f.dt >= 0 ? load_unaligned<int32_t>(row + f.dt) : 1;
    out_board[r] = f.board >= 0 ? load_unaligned<int16_t>(row + f.board) : (int16_t)0;
    out_chan[r] = f.chan >= 0 ? load_unaligned<int16_t>(row + f.chan) : (int16_t)0;
    out_rid[r] = rid;
    bm_bytes[r] = (int32_t)(((int64_t)(len > 0 ? len : 0) + 7 + 63) / 64 * 8 + 8);
    // the checks of the column route, first offender per kind (the host reports the earliest one)
    if (off < 0) atomicMin(&res->first_bad[0], (unsigned long long)r);
    if (len < 0) atomicMin(&res->first_bad[1], (unsigned long long)r);
    if (off >= 0 && len >= 0 && off + (int64_t)len > pool_n) atomicMin(&res->first_bad[2], (unsigned long long)r);
    if (len > WFA_MAX_RECORD_SAMPLES) atomicMin(&res->first_bad[3], (unsigned long long)r);
    if (len > 0) atomicMax(&res->max_len, len);
    // uniform layout (span mode / padded shadow): against record 0, read again by every thread (one cached line)
    const int64_t off0 = load_unaligned<int64_t>(rows + f.off);
    const int32_t len0 = load_unaligned<int32_t>(rows + f.len);
    const int8_t pol0 = pol_in ? pol_in[0] : pol_of(rows);
    if (len != len0 || off != off0 + r * (int64_t)len0 || (pol == WFA_POL_POSITIVE) != (pol0 == WFA_POL_POSITIVE))
        atomicOr(&res->nonuniform, 1);
    if (r > 0 && rid <= load_unaligned<int64_t>(row - f.row_bytes + f.rid)) atomicOr(&res->rid_not_increasing, 1);
}


extern "C" {

int wfa_abi_version(void) { return WFA_ABI_VERSION; }

int wfa_device_count(int* count) {
    if (!count) return fail(WFA_E_INVALID, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(WFA_E_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return WFA_OK;
}

int wfa_last_error(char* buf, size_t buf_len) {
    if (!buf || buf_len == 0) return WFA_E_INVALID;
    snprintf(buf, buf_len, "%s", g_last_error.c_str());
    return WFA_OK;
}

int wfa_ctx_create(int device_id, wfa_ctx** out) {
    if (!out) return fail(WFA_E_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    WFA_HIP_CHECK(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n)
        return fail(WFA_E_INVALID, "device %d out of range (have %d)", device_id, n);
    WFA_HIP_CHECK(hipSetDevice(device_id));
    wfa_ctx* c = new (std::nothrow) wfa_ctx();
    if (!c) return fail(WFA_E_NOMEM, "out of host memory");
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->h_total), 4 * sizeof(int64_t) + sizeof(wfa::RunsCold), hipHostMallocDefault);
    if (e == hipSuccess) c->h_cold = reinterpret_cast<wfa::RunsCold*>(c->h_total + 4);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        wfa_ctx_destroy(c);
        return fail(WFA_E_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return WFA_OK;
}

void wfa_ctx_destroy(wfa_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)wfa_rccl_destroy(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->pool_u16, &c->pool_f32, &c->off, &c->len, &c->baseline, &c->pol, &c->thr,
                      &c->ts, &c->dt, &c->board, &c->chan, &c->rid, &c->fixed_bl, &c->bm_off, &c->bitmap,
                      &c->hit_desc, &c->bw_scratch, &c->peak_out, &c->peak_cand_n, &c->peak_cand_pos, &c->peak_cand_val, &c->peak_slot_pos, &c->peak_slot_val,
                        &c->peak_cand_state, &c->peak_cand_rec, &c->peak_accept, &c->peak_ips, &c->peak_row_start, &c->wh_pos, &c->wh_row, &c->wh_valid, &c->sg.tab,
                      &c->sg.itab, &c->sg.sym, &c->hit_tmp, &c->cursor, &c->rec_tmp_start,
                      &c->rec_nhits, &c->rec_out_start, &c->scan_blocks, &c->hit_out, &c->out_rows, &c->out_rows2,
                      &c->gathered, &c->pw_plan, &c->fw_ties, &c->run_ev, &c->run_span_off, &c->run_span_cnt, &c->run_span_row0, &c->run_scan_blocks, &c->run_ctrl, &c->run_groups, &c->run_lit,
                      &c->shadow_pool, &c->shadow_off};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : c->ht) b.release();
    if (c->h_total) (void)hipHostFree(c->h_total);
    for (int b = 0; b < 2; ++b) {
        if (c->stage[b]) (void)hipHostFree(c->stage[b]);
        if (c->stage_ev[b]) (void)hipEventDestroy(c->stage_ev[b]);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    (void)profile_flush(c);
    for (hipEvent_t e : c->prof_free) (void)hipEventDestroy(e);
    c->prof_free.clear();
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int wfa_release_scratch(wfa_ctx* c, int64_t* freed_bytes) {
    if (!c) return fail(WFA_E_INVALID, "null context");
    WFA_HIP_CHECK(hipSetDevice(c->device));
    if (c->stream) WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    // what a later call rebuilds by itself.  Kept: the resident pools, the records columns, the filter plan, the rows of the
    // last passes (hit_out / hit_desc / peak_out / out_rows / gathered: *_fill and the resident hit-table stages read them),
    // the decoded CSV samples (source of wfa_pool_gather), the pinned staging ring and the small control blocks.
    DevBuf* bufs[] = {&c->bitmap, &c->hit_tmp, &c->bw_scratch, &c->peak_cand_n, &c->peak_cand_pos, &c->peak_cand_val,
                      &c->peak_slot_pos, &c->peak_slot_val, &c->peak_cand_state, &c->peak_cand_rec, &c->peak_accept,
                      &c->peak_ips, &c->peak_row_start, &c->wh_pos, &c->wh_row, &c->wh_valid, &c->fw_ties, &c->run_ev,
                      &c->shadow_pool, &c->shadow_off};
    int64_t freed = 0;
    for (DevBuf* b : bufs) { freed += (int64_t)b->cap; b->release(); }
    for (DevBuf& b : c->ht) { freed += (int64_t)b.cap; b.release(); }  // (a count pass without its fill is void after this)
    c->ht_n = -1;
    c->ht_perm = nullptr;
    c->bitmap_clean = false;
    c->shadow_valid = false;
    c->hit_tmp_rows = 0;
    if (freed_bytes) *freed_bytes = freed;
    return WFA_OK;
}

int wfa_set_option(wfa_ctx* c, const char* name, int value) {
    if (!c || !name) return fail(WFA_E_INVALID, "null argument");
    const std::string n(name);
    const bool v = value != 0;
    if (n == "no_fast") c->opt.no_fast = v;
    else if (n == "no_span") c->opt.no_span = v;
    else if (n == "no_pad") c->opt.no_pad = v;
    else if (n == "no_runs32") c->opt.no_runs32 = v;
    else if (n == "no_speculate") c->opt.no_speculate = v;
    else if (n == "no_peak_slots") c->opt.no_peak_slots = v;
    else if (n == "no_peak_hot") c->opt.no_peak_hot = v;
    else if (n == "no_deposit") c->opt.no_deposit = v;
    else if (n == "rows_grouped") c->opt.rows_grouped = v;
    else if (n == "span_records") c->opt.span_records = value;  // streaming kernel: records per span (0 = chosen by the library)
    else return fail(WFA_E_INVALID, "unknown option '%s'", name);
    return WFA_OK;
}

int wfa_last_h2d_rate(wfa_ctx* c, double* gb_per_s) {
    if (!c || !gb_per_s) return fail(WFA_E_INVALID, "null argument");
    *gb_per_s = c->last_h2d_GBps;
    return WFA_OK;
}

int wfa_sync(wfa_ctx* c) {
    int rc = use_device(c);
    if (rc) return rc;
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_upload_pool_u16(wfa_ctx* c, const uint16_t* pool, int64_t n) {
    int rc = use_device(c);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !pool)) return fail(WFA_E_INVALID, "bad wave_pool argument");
    if ((rc = h2d(c, c->pool_u16, pool, (size_t)n * sizeof(uint16_t)))) return rc;
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->pool_n = n;
    c->have_u16 = true;
    c->have_f32 = false;  // a filtered pool belongs to the previous wave_pool
    c->filter_keep = false;
    c->have_records = false;
    c->shadow_valid = false;
    c->pad_ok = false;
    return WFA_OK;
}

int wfa_upload_pool_f32(wfa_ctx* c, const float* pool, int64_t n) {
    int rc = use_device(c);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !pool)) return fail(WFA_E_INVALID, "bad wave_pool_filtered argument");
    if (c->have_u16 && n != c->pool_n) c->have_u16 = false;  // a different run: the resident wave_pool is stale
    if ((rc = h2d(c, c->pool_f32, pool, (size_t)n * sizeof(float)))) return rc;
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (!c->have_u16) {
        c->pool_n = n;
        c->have_records = false;
    }
    c->have_f32 = true;
    return WFA_OK;
}

int wfa_upload_records_soa(wfa_ctx* c, int64_t R, const int64_t* off, const int32_t* len,
                           const double* baseline, const int8_t* pol, const double* thr,
                           const int64_t* ts, const int32_t* dt, const int16_t* board,
                           const int16_t* chan, const int64_t* rid) {
    int rc = use_device(c);
    if (rc) return rc;
    if (!c->have_u16 && !c->have_f32) return fail(WFA_E_STATE, "upload a pool before the records");
    if (R < 0) return fail(WFA_E_INVALID, "negative record count");
    if (R > 0 && (!off || !len || !baseline || !pol || !thr || !ts || !dt || !board || !chan || !rid))
        return fail(WFA_E_INVALID, "null records column");
    int32_t max_len = 0;
    for (int64_t r = 0; r < R; ++r) {
        // same checks, same wording as data/records_view.py:47-56
        if (off[r] < 0) return fail(WFA_E_INVALID, "records contain negative wave_offset values");
        if (len[r] < 0) return fail(WFA_E_INVALID, "records contain negative event_length values");
        if (off[r] + (int64_t)len[r] > c->pool_n)
            return fail(WFA_E_INVALID, "records reference samples outside wave_pool bounds");
        if (len[r] > WFA_MAX_RECORD_SAMPLES)
            return fail(WFA_E_LIMIT, "record %lld has %d samples; this build supports at most %d",
                        (long long)r, len[r], WFA_MAX_RECORD_SAMPLES);
        if (pol[r] < WFA_POL_UNKNOWN || pol[r] > WFA_POL_POSITIVE_WAVE)
            return fail(WFA_E_INVALID, "bad polarity code %d at record %lld", (int)pol[r], (long long)r);
        if (len[r] > max_len) max_len = len[r];
    }
    // per-record region of the hit bitmap: bits indexed from the 16-byte aligned chunk base,
    // whole 64-bit words, one spare word
    std::vector<int64_t> bm_off((size_t)R);
    int64_t bm_total = 0;
    for (int64_t r = 0; r < R; ++r) {
        bm_off[(size_t)r] = bm_total;
        bm_total += ((int64_t)len[r] + 7 + 63) / 64 * 8 + 8;
    }
    c->bitmap_bytes = bm_total + 64;
    c->bitmap_clean = false;
    // span mode: every record the same length (multiple of 8, >= 24), laid out back to back from a
    // 16-byte aligned start, one polarity class
    c->span_ok = R > 0 && len[0] >= 24 && (len[0] % 8) == 0 && (off[0] % 8) == 0;
    for (int64_t r = 0; c->span_ok && r < R; ++r)
        c->span_ok = len[r] == len[0] && off[r] == off[0] + r * (int64_t)len[0] &&
                     (pol[r] == WFA_POL_POSITIVE) == (pol[0] == WFA_POL_POSITIVE);
    if (c->span_ok) {
        c->span_L = len[0];
        c->span_off0 = off[0];
        c->span_positive = pol[0] == WFA_POL_POSITIVE;
    }
    // padded shadow layout: uniform, back-to-back records whose length is not a multiple of 16 samples (VX2730: 1500)
    c->shadow_valid = false;
    c->pad_ok = R > 0 && c->have_u16 && len[0] >= 32 && (len[0] % 16) != 0;
    for (int64_t r = 0; c->pad_ok && r < R; ++r)
        c->pad_ok = len[r] == len[0] && off[r] == off[0] + r * (int64_t)len[0] &&
                    (pol[r] == WFA_POL_POSITIVE) == (pol[0] == WFA_POL_POSITIVE);
    if (c->pad_ok) {
        c->pad_L = len[0];
        c->pad_S = (len[0] + 15) / 16 * 16;
        c->pad_off0 = off[0];
        c->pad_positive = pol[0] == WFA_POL_POSITIVE;
    }
    const size_t n = (size_t)R;
    if ((rc = h2d(c, c->bm_off, bm_off.data(), n * 8))) return rc;
    if ((rc = h2d(c, c->off, off, n * 8))) return rc;
    if ((rc = h2d(c, c->len, len, n * 4))) return rc;
    if ((rc = h2d(c, c->baseline, baseline, n * 8))) return rc;
    if ((rc = h2d(c, c->pol, pol, n))) return rc;
    if ((rc = h2d(c, c->thr, thr, n * 8))) return rc;
    if ((rc = h2d(c, c->ts, ts, n * 8))) return rc;
    if ((rc = h2d(c, c->dt, dt, n * 4))) return rc;
    if ((rc = h2d(c, c->board, board, n * 2))) return rc;
    if ((rc = h2d(c, c->chan, chan, n * 2))) return rc;
    if ((rc = h2d(c, c->rid, rid, n * 8))) return rc;
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->R = R;
    c->max_len = max_len;
    c->pw_plan_n = -1;
    c->no_runs32 = false;
    c->have_records = true;
    c->n_hits = -1;
    return WFA_OK;
}


int wfa_upload_records_packed(wfa_ctx* c, const void* rows, int64_t R, int32_t row_bytes, const int32_t* field_offsets,
                              int32_t polarity_chars, double threshold, const double* thresholds, const int8_t* polarity,
                              int32_t* max_len_out, int* record_ids_increasing) {
    int rc = use_device(c);
    if (rc) return rc;
    if (!c->have_u16 && !c->have_f32) return fail(WFA_E_STATE, "upload a pool before the records");
    if (R < 0) return fail(WFA_E_INVALID, "negative record count");
    if (row_bytes <= 0 || !field_offsets || (R > 0 && !rows)) return fail(WFA_E_INVALID, "bad packed rows");
    PackedFields f{};
    int32_t* fo = &f.off;
    // order: wave_offset, event_length, baseline, polarity, timestamp, dt, board, channel, record_id
    const int32_t width[9] = {8, 4, 8, 4 * polarity_chars, 8, 4, 2, 2, 8};
    const bool required[9] = {true, true, true, false, true, false, false, false, true};
    for (int k = 0; k < 9; ++k) {
        fo[k] = field_offsets[k];
        if (fo[k] < 0 && required[k]) return fail(WFA_E_INVALID, "packed rows lack a required field (index %d)", k);
        if (fo[k] >= 0 && fo[k] + width[k] > row_bytes) return fail(WFA_E_INVALID, "field %d does not fit a %d-byte row", k, row_bytes);
    }
    if (f.polarity >= 0 && polarity_chars < 8) f.polarity = -1;  // neither "negative" nor "positive" fits: always unknown
    f.polarity_chars = polarity_chars;
    f.row_bytes = row_bytes;
    const size_t n = (size_t)R;
    if ((rc = c->off.ensure(n * 8)) || (rc = c->len.ensure(n * 4)) || (rc = c->baseline.ensure(n * 8)) || (rc = c->pol.ensure(n)) ||
        (rc = c->thr.ensure(n * 8)) || (rc = c->ts.ensure(n * 8)) || (rc = c->dt.ensure(n * 4)) || (rc = c->board.ensure(n * 2)) ||
        (rc = c->chan.ensure(n * 2)) || (rc = c->rid.ensure(n * 8)) || (rc = c->bm_off.ensure(n * 8)))
        return rc;
    c->have_records = false;
    c->R = 0;
    int32_t max_len = 0;
    bool increasing = true;
    c->span_ok = c->pad_ok = false;
    c->shadow_valid = false;
    c->bitmap_bytes = 64;
    if (R > 0) {
        // scratch: the rows, per-record bitmap bytes, the optional per-record columns of the caller, the result block
        DevBuf d_rows, d_bm, d_thr, d_pol, d_res, d_blocks;
        const int64_t nb = scan_blocks_for(R);
        if ((rc = h2d(c, d_rows, rows, n * (size_t)row_bytes))) return rc;
        if ((rc = d_bm.ensure(n * 4)) || (rc = d_res.ensure(sizeof(UnpackResult))) || (rc = d_blocks.ensure((size_t)(nb + 1) * 8))) return rc;
        if (thresholds && (rc = h2d(c, d_thr, thresholds, n * 8))) return rc;
        if (polarity && (rc = h2d(c, d_pol, polarity, n))) return rc;
        UnpackResult init{};
        for (auto& x : init.first_bad) x = ~0ull;
        WFA_HIP_CHECK(hipMemcpyAsync(d_res.ptr, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
        RecView cols{};
        cols.baseline_rw = c->baseline.as<double>();
        {
            LaunchTimer t(c);
            hipLaunchKernelGGL(k_unpack_records, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, c->stream,
                               d_rows.as<uint8_t>(), R, f, c->pool_n, threshold, thresholds ? d_thr.as<double>() : nullptr,
                               polarity ? d_pol.as<int8_t>() : nullptr, cols, c->pol.as<int8_t>(), c->thr.as<double>(),
                               c->off.as<int64_t>(), c->len.as<int32_t>(), c->ts.as<int64_t>(), c->dt.as<int32_t>(),
                               c->board.as<int16_t>(), c->chan.as<int16_t>(), c->rid.as<int64_t>(), d_bm.as<int32_t>(),
                               d_res.as<UnpackResult>());
            WFA_HIP_CHECK(hipGetLastError());
            WFA_HIP_CHECK(launch_scan(c->stream, d_bm.as<int32_t>(), R, d_blocks.as<int64_t>(), c->bm_off.as<int64_t>()));
            if ((rc = t.end("k_unpack_records + bitmap offsets"))) return rc;
        }
        UnpackResult res{};
        int64_t bm_total = 0;
        int64_t first[2] = {0, 0};
        int32_t len0 = 0;
        int8_t pol0 = 0;
        WFA_HIP_CHECK(hipMemcpyAsync(&res, d_res.ptr, sizeof(res), hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipMemcpyAsync(&bm_total, d_blocks.as<int64_t>() + nb, 8, hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipMemcpyAsync(&first[0], c->off.ptr, 8, hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipMemcpyAsync(&len0, c->len.ptr, 4, hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipMemcpyAsync(&pol0, c->pol.ptr, 1, hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        // the earliest offending record decides, its checks in the order of the column route (records_view.py:47-56)
        unsigned long long worst = ~0ull;
        for (auto x : res.first_bad) worst = x < worst ? x : worst;
        if (worst != ~0ull) {
            if (res.first_bad[0] == worst) return fail(WFA_E_INVALID, "records contain negative wave_offset values");
            if (res.first_bad[1] == worst) return fail(WFA_E_INVALID, "records contain negative event_length values");
            if (res.first_bad[2] == worst) return fail(WFA_E_INVALID, "records reference samples outside wave_pool bounds");
            return fail(WFA_E_LIMIT, "record %lld has more samples than this build supports (at most %d)", (long long)worst,
                        WFA_MAX_RECORD_SAMPLES);
        }
        if (polarity) {  // caller's codes: range-checked like the column route (the kernel only compared them)
            for (int64_t r = 0; r < R; ++r)
                if (polarity[r] < WFA_POL_UNKNOWN || polarity[r] > WFA_POL_POSITIVE_WAVE)
                    return fail(WFA_E_INVALID, "bad polarity code %d at record %lld", (int)polarity[r], (long long)r);
        }
        max_len = res.max_len;
        increasing = res.rid_not_increasing == 0;
        c->bitmap_bytes = bm_total + 64;
        const bool uniform = res.nonuniform == 0;
        c->span_ok = uniform && len0 >= 24 && (len0 % 8) == 0 && (first[0] % 8) == 0;
        if (c->span_ok) { c->span_L = len0; c->span_off0 = first[0]; c->span_positive = pol0 == WFA_POL_POSITIVE; }
        c->pad_ok = uniform && c->have_u16 && len0 >= 32 && (len0 % 16) != 0;
        if (c->pad_ok) { c->pad_L = len0; c->pad_S = (len0 + 15) / 16 * 16; c->pad_off0 = first[0]; c->pad_positive = pol0 == WFA_POL_POSITIVE; }
    }
    c->bitmap_clean = false;
    c->R = R;
    c->max_len = max_len;
    c->pw_plan_n = -1;
    c->no_runs32 = false;
    c->have_records = true;
    c->n_hits = -1;
    if (max_len_out) *max_len_out = max_len;
    if (record_ids_increasing) *record_ids_increasing = increasing ? 1 : 0;
    return WFA_OK;
}

int wfa_set_sg_plan(wfa_ctx* c, int window, int polyorder, const double* tab, const uint8_t* symmetric,
                    int int_ok, const int32_t* itab, int32_t den, int32_t den_edge, int64_t guard,
                    int64_t guard_edge) {
    int rc = use_device(c);
    if (rc) return rc;
    if (window < 1 || (window & 1) == 0 || window > WFA_MAX_SG_WINDOW)
        return fail(WFA_E_INVALID, "Savitzky-Golay window must be odd and in [1, %d], got %d",
                    WFA_MAX_SG_WINDOW, window);
    if (polyorder < 0 || polyorder >= window)
        return fail(WFA_E_INVALID, "polyorder %d must be in [0, window)", polyorder);
    if (!tab || !symmetric) return fail(WFA_E_INVALID, "null plan table");
    if (int_ok && (!itab || den <= 0 || den_edge <= 0)) return fail(WFA_E_INVALID, "bad integer plan");
    SgPlanDev& s = c->sg;
    s.W = window; s.P = polyorder; s.H = window / 2;
    s.n_tables = (window + 1) / 2;
    s.stride = window + 2 * s.H * window;
    s.int_ok = int_ok ? 1 : 0;
    s.den = int_ok ? den : 1;
    s.den_edge = int_ok ? den_edge : 1;
    s.guard = guard; s.guard_edge = guard_edge;
    s.rden = 1.0 / (double)s.den;
    s.rden_edge = 1.0 / (double)s.den_edge;
    if ((rc = h2d(c, s.tab, tab, (size_t)s.n_tables * s.stride * sizeof(double)))) return rc;
    if ((rc = h2d(c, s.sym, symmetric, (size_t)s.n_tables))) return rc;
    const size_t isz = (size_t)s.stride * sizeof(int32_t);
    if (int_ok) {
        if ((rc = h2d(c, s.itab, itab, isz))) return rc;
    } else {
        if ((rc = s.itab.ensure(isz))) return rc;
    }
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->have_sg = true;
    return WFA_OK;
}

int wfa_baseline_mean(wfa_ctx* c, int32_t start, int32_t end, int update_records, double* out) {
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = need_source(c, WFA_SRC_RAW))) return rc;
    if (start < 0) return fail(WFA_E_INVALID, "baseline window start must be >= 0");
    if (c->R == 0) return WFA_OK;
    double* dst = c->baseline.as<double>();
    if (!update_records) {
        if ((rc = c->out_rows.ensure((size_t)c->R * sizeof(double)))) return rc;
        dst = c->out_rows.as<double>();
    }
    {
        LaunchTimer t(c);
        WFA_HIP_CHECK(launch_baseline_mean(c->stream, pool_view(c), rec_view(c), start, end, dst));
        if ((rc = t.end("k_baseline_mean"))) return rc;
    }
    if (out)
        WFA_HIP_CHECK(hipMemcpyAsync(out, dst, (size_t)c->R * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_filter_keep_output(wfa_ctx* c, int keep) {
    int rc = use_device(c);
    if (rc) return rc;
    if (keep && (!c->have_f32 || c->pool_f32.cap < (size_t)c->pool_n * sizeof(float)))
        return fail(WFA_E_STATE, "keep requested but no filtered output exists yet");
    c->filter_keep = keep != 0;
    return WFA_OK;
}

int wfa_download_pool_f32(wfa_ctx* c, float* out, int64_t n) {
    int rc = use_device(c);
    if (rc) return rc;
    if (!c->have_f32) return fail(WFA_E_STATE, "no float32 pool is resident");
    if (n != c->pool_n) return fail(WFA_E_INVALID, "caller expects %lld samples, the pool has %lld", (long long)n, (long long)c->pool_n);
    if (n == 0) return WFA_OK;
    if (!out) return fail(WFA_E_INVALID, "out is null");
    WFA_HIP_CHECK(hipMemcpyAsync(out, c->pool_f32.ptr, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_savgol(wfa_ctx* c, float* out) {
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = need_source(c, WFA_SRC_SG_FUSED))) return rc;
    if ((rc = c->pool_f32.ensure((size_t)c->pool_n * sizeof(float)))) return rc;
    // gaps between records stay 0.0 (records.py:382)
    if (!c->filter_keep) WFA_HIP_CHECK(hipMemsetAsync(c->pool_f32.ptr, 0, (size_t)c->pool_n * sizeof(float), c->stream));
    if (c->R > 0) {
        PoolView pv = pool_view(c);
        const SgParams sp0 = sg_params(c);
        const bool padded = c->pad_ok && c->pad_L >= sp0.W && sg_mask_supported(sp0) && !c->opt.no_fast &&
                            !c->opt.no_pad;
        if (padded && (rc = ensure_shadow(c))) return rc;
        LaunchTimer t(c);
        if (padded) {  // uniform records, L % 16 != 0: span kernel reading the padded shadow, writing the packed pool
            SpanParams sp{};
            sp.off0 = 0; sp.L = c->pad_L; sp.S = c->pad_S; sp.out_off0 = c->pad_off0; sp.positive = 0;
            sp.rs = 64;
            sp.n_spans = (c->R + sp.rs - 1) / sp.rs;
            pv.u16 = c->shadow_pool.as<uint16_t>();
            pv.n = c->R * (int64_t)c->pad_S;
            WFA_HIP_CHECK(launch_savgol_span(c->stream, pv, rec_view(c), sp0, sp, c->pool_f32.as<float>()));
            if ((rc = t.end("k_savgol_span<padded>"))) return rc;
        } else if (c->span_ok && c->span_L >= 16 && c->span_L >= sp0.W && sg_mask_supported(sp0) &&
            !c->opt.no_fast) {
            SpanParams sp{};
            sp.off0 = c->span_off0; sp.L = c->span_L; sp.positive = 0;
            sp.rs = 64;
            sp.n_spans = (c->R + sp.rs - 1) / sp.rs;
            WFA_HIP_CHECK(launch_savgol_span(c->stream, pv, rec_view(c), sp0, sp, c->pool_f32.as<float>()));
            if ((rc = t.end("k_savgol_span"))) return rc;
        } else {
            WFA_HIP_CHECK(launch_savgol(c->stream, pv, rec_view(c), sp0, c->pool_f32.as<float>()));
            if ((rc = t.end("k_savgol"))) return rc;
        }
    }
    c->have_f32 = true;
    if (out)
        WFA_HIP_CHECK(hipMemcpyAsync(out, c->pool_f32.ptr, (size_t)c->pool_n * sizeof(float),
                                     hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_sosfiltfilt(wfa_ctx* c, int n_sections, const double* sos, const double* zi, int32_t padlen, float* out) {
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = need_source(c, WFA_SRC_RAW))) return rc;
    if (n_sections < 1 || n_sections > 8) return fail(WFA_E_INVALID, "n_sections must be in [1, 8], got %d", n_sections);
    if (!sos || !zi || padlen < 0) return fail(WFA_E_INVALID, "bad sosfiltfilt arguments");
    if ((rc = c->pool_f32.ensure((size_t)c->pool_n * sizeof(float)))) return rc;
    if (!c->filter_keep) WFA_HIP_CHECK(hipMemsetAsync(c->pool_f32.ptr, 0, (size_t)c->pool_n * sizeof(float), c->stream));
    if (c->R > 0) {
        // forward-pass scratch: [max_len + 2 padlen][batch] float64, batches bounded to ~2 GiB
        const int64_t n_ext = (int64_t)c->max_len + 2 * (int64_t)padlen;
        int64_t batch = (int64_t)(2147483648LL / (n_ext * 8));
        batch = batch / 256 * 256;
        if (batch < 256) batch = 256;
        if (batch > c->R) batch = (c->R + 255) / 256 * 256;
        if ((rc = c->bw_scratch.ensure((size_t)(n_ext * batch * 8)))) return rc;
        const PoolView pv = pool_view(c);
        const RecView rv = rec_view(c);
        LaunchTimer t(c);
        for (int64_t r0 = 0; r0 < c->R; r0 += batch) {
            const int64_t r1 = r0 + batch < c->R ? r0 + batch : c->R;
            WFA_HIP_CHECK(launch_sosfiltfilt(c->stream, pv, rv, n_sections, sos, zi, padlen, r0, r1,
                                             c->bw_scratch.as<double>(), batch, c->pool_f32.as<float>()));
        }
        if ((rc = t.end("k_sosfiltfilt"))) return rc;
    }
    c->have_f32 = true;
    if (out)
        WFA_HIP_CHECK(hipMemcpyAsync(out, c->pool_f32.ptr, (size_t)c->pool_n * sizeof(float), hipMemcpyDeviceToHost,
                                     c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_threshold_hits_count(wfa_ctx* c, int source, int32_t le, int32_t re, int32_t max_len,
                             int64_t* n_hits) {
    return run_hits(c, source, false, 0, 0, le, re, max_len, n_hits);
}

int wfa_fused_baseline_filter_hits(wfa_ctx* c, int32_t bl_start, int32_t bl_end, int32_t le, int32_t re,
                                   int32_t max_len, int64_t* n_hits) {
    return run_hits(c, WFA_SRC_SG_FUSED, bl_end > bl_start, bl_start, bl_end, le, re, max_len, n_hits);
}

int wfa_hits_enqueue(wfa_ctx* c, int source, int32_t bl_start, int32_t bl_end, int32_t le, int32_t re, int32_t max_len) {
    return run_hits(c, source, source == WFA_SRC_SG_FUSED && bl_end > bl_start, bl_start, bl_end, le, re, max_len,
                    nullptr, true);
}

int wfa_hits_wait(wfa_ctx* c, int64_t* n_hits) {
    int rc = use_device(c);
    if (rc) return rc;
    if (!n_hits) return fail(WFA_E_INVALID, "n_hits is null");
    if (c->pending) {
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->pending = false;
        const int64_t total = *c->h_total;
        const bool runs_bad = c->pend.runs32 && (c->h_total[2] & 0xffffffffll) != 0;  // event buffers overflowed
        if (runs_bad) c->no_runs32 = true;  // a span outgrew its LDS buffer or its slot: the general route from now on
        if (total <= c->pend.bound && !runs_bad) {
            c->last_hits = total;
            c->n_hits = total;
        } else {  // more rows than the speculative launch covered: the exact route, now
            c->last_hits = total;
            const auto p = c->pend;
            if ((rc = run_hits(c, p.source, p.fused_bl, p.bl_start, p.bl_end, p.le, p.re, p.max_len, n_hits))) return rc;
            return WFA_OK;
        }
    }
    if (c->n_hits < 0) return fail(WFA_E_STATE, "no hit pass has been run");
    *n_hits = c->n_hits;
    return WFA_OK;
}

int wfa_threshold_hits_fill(wfa_ctx* c, void* out_rows, int64_t n_hits) {
    int rc = use_device(c);
    if (rc) return rc;
    if (c->pending) {
        int64_t n = 0;
        if ((rc = wfa_hits_wait(c, &n))) return rc;
    }
    if (c->n_hits < 0) return fail(WFA_E_STATE, "no hit pass has been run");
    if (n_hits != c->n_hits)
        return fail(WFA_E_INVALID, "caller expects %lld rows, the pass produced %lld", (long long)n_hits,
                    (long long)c->n_hits);
    if (n_hits == 0) return WFA_OK;
    if (!out_rows) return fail(WFA_E_INVALID, "out_rows is null");
    WFA_HIP_CHECK(hipMemcpyAsync(out_rows, c->hit_out.ptr, (size_t)n_hits * 60, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_find_peaks_count(wfa_ctx* c, int source, int signal_mode, int use_derivative, double height, int has_threshold,
                         double threshold, int32_t distance, double prominence, double width, int height_method,
                         int32_t ext, int64_t* n_peaks) {
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = need_source(c, source))) return rc;
    if (source == WFA_SRC_SG_FUSED) return fail(WFA_E_INVALID, "find_peaks reads wave_pool or a materialised wave_pool_filtered");
    if (!n_peaks) return fail(WFA_E_INVALID, "n_peaks is null");
    if (distance < 1) return fail(WFA_E_INVALID, "`distance` must be greater or equal to 1");  // scipy's message
    if (height_method != WFA_HEIGHT_MINMAX && height_method != WFA_HEIGHT_DIFF)
        return fail(WFA_E_INVALID, "height_method must be WFA_HEIGHT_MINMAX or WFA_HEIGHT_DIFF");
    if (signal_mode < WFA_PEAK_SIGNAL_RECORDS || signal_mode > WFA_PEAK_SIGNAL_ROWS_F64)
        return fail(WFA_E_INVALID, "signal_mode must be WFA_PEAK_SIGNAL_RECORDS, _ROWS or _ROWS_F64");
    c->n_peaks = -1;
    if (c->R == 0) { c->n_peaks = 0; *n_peaks = 0; return WFA_OK; }
    const int64_t R = c->R;
    if ((rc = c->rec_nhits.ensure(R * sizeof(int32_t)))) return rc;
    if ((rc = c->rec_out_start.ensure(R * sizeof(int64_t)))) return rc;
    const int64_t nb = scan_blocks_for(R);
    if ((rc = c->scan_blocks.ensure((nb + 1) * sizeof(int64_t)))) return rc;
    if ((rc = c->cursor.ensure(sizeof(unsigned long long)))) return rc;
    WFA_HIP_CHECK(hipMemsetAsync(c->cursor.ptr, 0, sizeof(unsigned long long), c->stream));
    PeakParams pp{};
    pp.use_derivative = use_derivative ? 1 : 0; pp.hmin = height; pp.has_threshold = has_threshold ? 1 : 0;
    pp.tmin = threshold; pp.distance = distance; pp.pmin = prominence; pp.wmin = width;
    pp.ext = ext > 0 ? ext : 0; pp.height_diff = height_method == WFA_HEIGHT_DIFF;
    pp.rows = signal_mode;
    const PoolView pv = pool_view(c);
    const RecView rv = rec_view(c);
    int* err = reinterpret_cast<int*>(c->cursor.ptr);
    int32_t* counts = c->rec_nhits.as<int32_t>();
    auto scan_total = [&](int64_t* starts, int64_t* total) -> int {
        WFA_HIP_CHECK(launch_scan(c->stream, counts, R, c->scan_blocks.as<int64_t>(), starts));
        WFA_HIP_CHECK(hipMemcpyAsync(total, c->scan_blocks.as<int64_t>() + nb, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
        WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        return WFA_OK;
    };
    int64_t total = 0;
    {
        if ((rc = c->rec_tmp_start.ensure(R * sizeof(int64_t)))) return rc;
        if ((rc = c->peak_cand_n.ensure(R * sizeof(int32_t)))) return rc;
        int64_t* cand_start = c->rec_tmp_start.as<int64_t>();
        int32_t* cand_n = c->peak_cand_n.as<int32_t>();
        int64_t n_cand = 0;
        // one walk: counts + the first kPeakSlots candidates of every record in per-record slots (launch_find_peaks_slots);
        // `no_peak_slots` keeps the count + fill pair (two walks, any number of candidates)
        const int K = kPeakSlots;
        const bool slots = !c->opt.no_peak_slots;
        int* overflow = err + 1;
        if (slots) {
            if ((rc = c->peak_slot_pos.ensure((size_t)R * K * sizeof(int32_t)))) return rc;
            if ((rc = c->peak_slot_val.ensure((size_t)R * K * sizeof(double)))) return rc;
            LaunchTimer t(c);
            hipError_t herr = hipSuccess;
            // uniform records: the walk on LDS-staged groups (coalesced); any other layout: one lane per record
            // uniform records: LDS-staged groups (coalesced) -- the plateau machine only on the chunks that can hold a value
            // >= `height` (k_find_peaks_hot), or over every sample (k_find_peaks_staged)
            const bool hot = c->span_ok && !c->opt.no_span && !c->opt.no_peak_hot &&
                             launch_find_peaks_hot(c->stream, source, pv, rv, pp, c->span_off0, c->span_L, K, counts,
                                                   c->peak_slot_pos.as<int32_t>(), c->peak_slot_val.as<double>(), overflow, &herr);
            WFA_HIP_CHECK(herr);
            const bool staged = !hot && c->span_ok && !c->opt.no_span &&
                                launch_find_peaks_staged(c->stream, source, pv, rv, pp, c->span_off0, c->span_L, K, counts,
                                                         c->peak_slot_pos.as<int32_t>(), c->peak_slot_val.as<double>(), overflow, &herr);
            WFA_HIP_CHECK(herr);
            if (!staged && !hot)
                WFA_HIP_CHECK(launch_find_peaks_slots(c->stream, source, pv, rv, pp, K, counts, c->peak_slot_pos.as<int32_t>(),
                                                      c->peak_slot_val.as<double>(), overflow));
            if ((rc = t.end(hot ? "k_find_peaks_hot" : staged ? "k_find_peaks_staged" : "k_find_peaks_slots"))) return rc;
        } else {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_find_peaks(c->stream, source, false, pv, rv, pp, counts, nullptr, nullptr, nullptr, nullptr));
            if ((rc = t.end("k_find_peaks<count candidates>"))) return rc;
        }
        int over = 0;
        WFA_HIP_CHECK(hipMemcpyAsync(&over, overflow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if ((rc = scan_total(cand_start, &n_cand))) return rc;
        const size_t nc = (size_t)(n_cand > 0 ? n_cand : 1);
        if ((rc = c->peak_cand_pos.ensure(nc * sizeof(int32_t)))) return rc;
        if ((rc = c->peak_cand_val.ensure(nc * sizeof(double)))) return rc;
        if ((rc = c->peak_cand_rec.ensure(nc * sizeof(int64_t)))) return rc;
        if ((rc = c->peak_cand_state.ensure(nc))) return rc;
        if ((rc = c->peak_accept.ensure(nc * sizeof(int32_t)))) return rc;
        if ((rc = c->peak_ips.ensure(nc * 2 * sizeof(double)))) return rc;
        if ((rc = c->peak_row_start.ensure(nc * sizeof(int64_t)))) return rc;
        int32_t* cpos = c->peak_cand_pos.as<int32_t>();
        double* cval = c->peak_cand_val.as<double>();
        int64_t* crec = c->peak_cand_rec.as<int64_t>();
        uint8_t* cstate = distance > 2 ? c->peak_cand_state.as<uint8_t>() : nullptr;
        int32_t* accept = c->peak_accept.as<int32_t>();
        if (slots && !over) {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_peak_compact(c->stream, R, K, counts, cand_start, c->peak_slot_pos.as<int32_t>(),
                                              c->peak_slot_val.as<double>(), cpos, cval, crec));
            if ((rc = t.end("k_peak_compact"))) return rc;
        } else {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_find_peaks(c->stream, source, true, pv, rv, pp, nullptr, cand_start, cpos, cval, crec));
            if ((rc = t.end("k_find_peaks<fill candidates>"))) return rc;
        }
        if (distance > 2) {  // local maxima are at least 2 samples apart: scipy's distance step keeps all of them otherwise
            WFA_HIP_CHECK(hipMemcpyAsync(cand_n, counts, R * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_peak_select(c->stream, R, cand_n, cand_start, cpos, cval, cstate, distance));
            if ((rc = t.end("k_peak_select"))) return rc;
        }
        {
            LaunchTimer t(c);
            WFA_HIP_CHECK(launch_peak_eval(c->stream, source, pv, rv, pp, n_cand, crec, cpos, cstate, accept, c->peak_ips.as<double>()));
            if ((rc = t.end("k_peak_eval"))) return rc;
        }
        if (n_cand > 0) {
            const int64_t nbc = scan_blocks_for(n_cand);
            if ((rc = c->scan_blocks.ensure((nbc + 1) * sizeof(int64_t)))) return rc;
            WFA_HIP_CHECK(launch_scan(c->stream, accept, n_cand, c->scan_blocks.as<int64_t>(), c->peak_row_start.as<int64_t>()));
            WFA_HIP_CHECK(hipMemcpyAsync(&total, c->scan_blocks.as<int64_t>() + nbc, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
            WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        if ((rc = c->peak_out.ensure((size_t)total * 48))) return rc;
        LaunchTimer t(c);
        WFA_HIP_CHECK(launch_peak_rows(c->stream, source, pv, rv, pp, n_cand, crec, cpos, accept, c->peak_row_start.as<int64_t>(),
                                       c->peak_ips.as<double>(), c->peak_out.as<uint8_t>(), err));
        if ((rc = t.end("k_peak_rows"))) return rc;
    }
    int flag = 0;
    WFA_HIP_CHECK(hipMemcpyAsync(&flag, err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (flag)  // numpy's message for np.max of the empty height window (peak_finding.py:606)
        return fail(WFA_E_INVALID, "zero-size array to reduction operation maximum which has no identity");
    c->n_peaks = total;
    *n_peaks = total;
    return WFA_OK;
}

int wfa_find_peaks_fill(wfa_ctx* c, void* out_rows, int64_t n_peaks) {
    int rc = use_device(c);
    if (rc) return rc;
    if (c->n_peaks < 0) return fail(WFA_E_STATE, "no find_peaks pass has been run");
    if (n_peaks != c->n_peaks)
        return fail(WFA_E_INVALID, "caller expects %lld rows, the pass produced %lld", (long long)n_peaks, (long long)c->n_peaks);
    if (n_peaks == 0) return WFA_OK;
    if (!out_rows) return fail(WFA_E_INVALID, "out_rows is null");
    WFA_HIP_CHECK(hipMemcpyAsync(out_rows, c->peak_out.ptr, (size_t)n_peaks * 48, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_basic_features(wfa_ctx* c, int source, int64_t h0, int64_t h1, int h_has_end, int64_t a0,
                       int64_t a1, int a_has_end, const double* fixed_baseline, void* out_rows) {
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = need_source(c, source))) return rc;
    if (source == WFA_SRC_SG_FUSED)
        return fail(WFA_E_INVALID, "basic_features reads wave_pool or a materialised wave_pool_filtered");
    if (c->R == 0) return WFA_OK;
    if (!out_rows) return fail(WFA_E_INVALID, "out_rows is null");
    FeatParams fp{};
    fp.h0 = h0; fp.h1 = h1; fp.h_has_end = h_has_end;
    fp.a0 = a0; fp.a1 = a1; fp.a_has_end = a_has_end;
    fp.fixed_bl = nullptr;
    if (fixed_baseline) {
        if ((rc = h2d(c, c->fixed_bl, fixed_baseline, (size_t)c->R * sizeof(double)))) return rc;
        fp.fixed_bl = c->fixed_bl.as<double>();
    }
    if ((rc = c->out_rows.ensure((size_t)c->R * 36))) return rc;
    {
        LaunchTimer t(c);
        hipError_t e = hipSuccess;
        if (source == WFA_SRC_RAW && launch_basic_features_wave(c, rec_view(c), fp, c->out_rows.as<uint8_t>(), &e)) {
            WFA_HIP_CHECK(e);
            if ((rc = t.end("k_basic_features_leaf"))) return rc;
        } else {
            WFA_HIP_CHECK(launch_basic_features(c->stream, source, pool_view(c), rec_view(c), sg_params(c), fp,
                                                c->out_rows.as<uint8_t>()));
            if ((rc = t.end("k_basic_features"))) return rc;
        }
    }
    WFA_HIP_CHECK(hipMemcpyAsync(out_rows, c->out_rows.ptr, (size_t)c->R * 36, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_find_hits_count(wfa_ctx* c, int source, int64_t n_rows, int32_t row_length, const double* baselines,
                        double threshold, int64_t* n_hits) {
    int rc = use_device(c);
    if (rc) return rc;
    if (source != WFA_SRC_RAW && source != WFA_SRC_F32)
        return fail(WFA_E_INVALID, "find_hits reads dense rows: source must be WFA_SRC_RAW or WFA_SRC_F32");
    if (source == WFA_SRC_RAW ? !c->have_u16 : !c->have_f32) return fail(WFA_E_STATE, "no wave matrix uploaded for this source");
    if (n_rows < 0 || row_length < 0 || !n_hits) return fail(WFA_E_INVALID, "bad arguments");
    if (n_rows * (int64_t)row_length > c->pool_n)
        return fail(WFA_E_INVALID, "wave matrix %lld x %d exceeds the resident pool (%lld samples)", (long long)n_rows,
                    row_length, (long long)c->pool_n);
    c->n_legacy = -1;
    if (n_rows == 0 || row_length == 0) { c->n_legacy = 0; *n_hits = 0; return WFA_OK; }
    if (!baselines) return fail(WFA_E_INVALID, "baselines is null");
    if ((rc = c->fixed_bl.ensure((size_t)n_rows * sizeof(double)))) return rc;
    WFA_HIP_CHECK(hipMemcpyAsync(c->fixed_bl.ptr, baselines, (size_t)n_rows * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((rc = c->rec_nhits.ensure(n_rows * sizeof(int32_t)))) return rc;
    if ((rc = c->rec_out_start.ensure(n_rows * sizeof(int64_t)))) return rc;
    const int64_t nb = scan_blocks_for(n_rows);
    if ((rc = c->scan_blocks.ensure((nb + 1) * sizeof(int64_t)))) return rc;
    const PoolView pv = pool_view(c);
    LaunchTimer t(c);
    WFA_HIP_CHECK(launch_find_hits_legacy(c->stream, source, false, pv, n_rows, row_length, c->fixed_bl.as<double>(), threshold,
                                          c->rec_nhits.as<int32_t>(), nullptr, nullptr, nullptr));
    WFA_HIP_CHECK(launch_scan(c->stream, c->rec_nhits.as<int32_t>(), n_rows, c->scan_blocks.as<int64_t>(),
                              c->rec_out_start.as<int64_t>()));
    int64_t total = 0;
    WFA_HIP_CHECK(hipMemcpyAsync(&total, c->scan_blocks.as<int64_t>() + nb, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    if ((rc = c->peak_row_start.ensure((size_t)(total > 0 ? total : 1) * sizeof(int64_t)))) return rc;
    if ((rc = c->peak_ips.ensure((size_t)(total > 0 ? total : 1) * sizeof(int64_t)))) return rc;
    WFA_HIP_CHECK(launch_find_hits_legacy(c->stream, source, true, pv, n_rows, row_length, c->fixed_bl.as<double>(), threshold,
                                          nullptr, c->rec_out_start.as<int64_t>(), c->peak_row_start.as<int64_t>(),
                                          c->peak_ips.as<int64_t>()));
    if ((rc = t.end("k_find_hits_legacy (count + scan + fill)"))) return rc;
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->n_legacy = total;
    *n_hits = total;
    return WFA_OK;
}

int wfa_find_hits_fill(wfa_ctx* c, int64_t n_hits, int64_t* event_index, int64_t* start_sample) {
    int rc = use_device(c);
    if (rc) return rc;
    if (c->n_legacy < 0) return fail(WFA_E_STATE, "no find_hits pass has been run");
    if (n_hits != c->n_legacy)
        return fail(WFA_E_INVALID, "caller expects %lld hits, the pass produced %lld", (long long)n_hits, (long long)c->n_legacy);
    if (n_hits == 0) return WFA_OK;
    if (!event_index || !start_sample) return fail(WFA_E_INVALID, "null output");
    WFA_HIP_CHECK(hipMemcpyAsync(event_index, c->peak_row_start.ptr, (size_t)n_hits * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(start_sample, c->peak_ips.ptr, (size_t)n_hits * 8, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_waveform_width(wfa_ctx* c, int source, int64_t n_hits, const int64_t* position, const int64_t* row_index,
                       int64_t n_rows, int32_t row_length, double rise_low, double rise_high, double fall_high,
                       double fall_low, double sampling_rate, int interpolation, void* out_rows, uint8_t* valid) {
    int rc = use_device(c);
    if (rc) return rc;
    if (source != WFA_SRC_RAW && source != WFA_SRC_F32)
        return fail(WFA_E_INVALID, "waveform_width reads dense rows: source must be WFA_SRC_RAW or WFA_SRC_F32");
    if (source == WFA_SRC_RAW ? !c->have_u16 : !c->have_f32) return fail(WFA_E_STATE, "no wave matrix uploaded for this source");
    if (n_hits < 0 || n_rows < 0 || row_length < 0) return fail(WFA_E_INVALID, "negative size");
    if (n_rows * (int64_t)row_length > c->pool_n)
        return fail(WFA_E_INVALID, "wave matrix %lld x %d exceeds the resident pool (%lld samples)", (long long)n_rows,
                    row_length, (long long)c->pool_n);
    if (n_hits == 0) return WFA_OK;
    if (!position || !row_index || !out_rows || !valid) return fail(WFA_E_INVALID, "null argument");
    if (!(sampling_rate == sampling_rate) || sampling_rate == 0.0) return fail(WFA_E_INVALID, "float division by zero");
    // scratch: positions, row indices, rows, valid bytes
    const size_t b_idx = (size_t)n_hits * sizeof(int64_t);
    if ((rc = c->wh_pos.ensure(b_idx))) return rc;
    if ((rc = c->wh_row.ensure(b_idx))) return rc;
    if ((rc = c->out_rows.ensure((size_t)n_hits * 56))) return rc;
    if ((rc = c->wh_valid.ensure((size_t)n_hits))) return rc;
    WFA_HIP_CHECK(hipMemcpyAsync(c->wh_pos.ptr, position, b_idx, hipMemcpyHostToDevice, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(c->wh_row.ptr, row_index, b_idx, hipMemcpyHostToDevice, c->stream));
    {
        LaunchTimer t(c);
        WFA_HIP_CHECK(launch_waveform_width(c->stream, source, pool_view(c), n_hits, c->wh_pos.as<int64_t>(),
                                            c->wh_row.as<int64_t>(), n_rows, row_length, rise_low, rise_high, fall_high,
                                            fall_low, sampling_rate, interpolation, c->out_rows.as<uint8_t>(),
                                            c->wh_valid.as<uint8_t>()));
        if ((rc = t.end("k_waveform_width"))) return rc;
    }
    WFA_HIP_CHECK(hipMemcpyAsync(out_rows, c->out_rows.ptr, (size_t)n_hits * 56, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipMemcpyAsync(valid, c->wh_valid.ptr, (size_t)n_hits, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_width_integral(wfa_ctx* c, int source, double q_low, double q_high, double dt, void* out_rows) {
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = need_source(c, source))) return rc;
    if (source == WFA_SRC_SG_FUSED)
        return fail(WFA_E_INVALID, "width_integral reads wave_pool or a materialised wave_pool_filtered");
    if (!(q_low > 0.0) || !(q_high < 1.0) || !(q_low < q_high))  // waveform_width_integral.py:95-96
        return fail(WFA_E_INVALID, "q_low/q_high invalid: q_low=%g, q_high=%g", q_low, q_high);
    if (c->R == 0) return WFA_OK;
    if (!out_rows) return fail(WFA_E_INVALID, "out_rows is null");
    WidthParams wp{q_low, q_high, dt};
    int rc2 = c->out_rows.ensure((size_t)c->R * 52);
    if (rc2) return rc2;
    {
        LaunchTimer t(c);
        hipError_t e = hipSuccess;
        if (source == WFA_SRC_RAW && launch_width_integral_wave(c, rec_view(c), wp, c->out_rows.as<uint8_t>(), &e)) {
            WFA_HIP_CHECK(e);
            if ((rc = t.end("k_width_integral_leaf"))) return rc;
        } else {
            WFA_HIP_CHECK(launch_width_integral(c->stream, source, pool_view(c), rec_view(c), sg_params(c), wp,
                                                c->out_rows.as<uint8_t>()));
            if ((rc = t.end("k_width_integral"))) return rc;
        }
    }
    WFA_HIP_CHECK(hipMemcpyAsync(out_rows, c->out_rows.ptr, (size_t)c->R * 52, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_features_both(wfa_ctx* c, int64_t h0, int64_t h1, int h_has_end, int64_t a0, int64_t a1, int a_has_end,
                      double q_low, double q_high, double dt, void* out_basic, void* out_width) {
    int rc = use_device(c);
    if (rc) return rc;
    if ((rc = need_source(c, WFA_SRC_RAW))) return rc;
    if (!(q_low > 0.0) || !(q_high < 1.0) || !(q_low < q_high))  // waveform_width_integral.py:95-96
        return fail(WFA_E_INVALID, "q_low/q_high invalid: q_low=%g, q_high=%g", q_low, q_high);
    if (c->R == 0) return WFA_OK;
    FeatParams fp{};
    fp.h0 = h0; fp.h1 = h1; fp.h_has_end = h_has_end;
    fp.a0 = a0; fp.a1 = a1; fp.a_has_end = a_has_end;
    fp.fixed_bl = nullptr;
    WidthParams wp{q_low, q_high, dt};
    if ((rc = c->out_rows.ensure((size_t)c->R * 36)) || (rc = c->out_rows2.ensure((size_t)c->R * 52))) return rc;
    uint8_t* ob = c->out_rows.as<uint8_t>();
    uint8_t* ow = c->out_rows2.as<uint8_t>();
    {
        LaunchTimer t(c);
        hipError_t e = hipSuccess;
        if (launch_features_both_wave(c, rec_view(c), fp, wp, ob, ow, &e)) {
            WFA_HIP_CHECK(e);
            if ((rc = t.end("k_features_both_leaf"))) return rc;
        } else {  // layout or ranges outside the fused kernel: the two kernels, one after the other
            if (launch_basic_features_wave(c, rec_view(c), fp, ob, &e)) {
                WFA_HIP_CHECK(e);
                if ((rc = t.end("k_basic_features_leaf"))) return rc;
            } else {
                WFA_HIP_CHECK(launch_basic_features(c->stream, WFA_SRC_RAW, pool_view(c), rec_view(c), sg_params(c), fp, ob));
                if ((rc = t.end("k_basic_features"))) return rc;
            }
            LaunchTimer t2(c);
            if (launch_width_integral_wave(c, rec_view(c), wp, ow, &e)) {
                WFA_HIP_CHECK(e);
                if ((rc = t2.end("k_width_integral_leaf"))) return rc;
            } else {
                WFA_HIP_CHECK(launch_width_integral(c->stream, WFA_SRC_RAW, pool_view(c), rec_view(c), sg_params(c), wp, ow));
                if ((rc = t2.end("k_width_integral"))) return rc;
            }
        }
    }
    if (out_basic) WFA_HIP_CHECK(hipMemcpyAsync(out_basic, ob, (size_t)c->R * 36, hipMemcpyDeviceToHost, c->stream));
    if (out_width) WFA_HIP_CHECK(hipMemcpyAsync(out_width, ow, (size_t)c->R * 52, hipMemcpyDeviceToHost, c->stream));
    WFA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return WFA_OK;
}

int wfa_profile_enable(wfa_ctx* c, int on) {
    if (!c) return fail(WFA_E_INVALID, "null context");
    if (!on) (void)profile_flush(c);
    c->prof_on = on != 0;
    c->prof_level = on == 2 ? 2 : (on ? 1 : 0);
    return WFA_OK;
}

int wfa_profile_reset(wfa_ctx* c) {
    if (!c) return fail(WFA_E_INVALID, "null context");
    (void)profile_flush(c);
    c->prof.clear();
    return WFA_OK;
}

int wfa_profile_get(wfa_ctx* c, int idx, char* name, size_t name_len, double* total_ms, int64_t* launches) {
    if (!c) return fail(WFA_E_INVALID, "null context");
    (void)hipSetDevice(c->device);
    (void)profile_flush(c);
    if (idx < 0 || idx >= (int)c->prof.size()) return WFA_E_INVALID;
    const ProfEntry& e = c->prof[idx];
    if (name && name_len) snprintf(name, name_len, "%s", e.name.c_str());
    if (total_ms) *total_ms = e.total_ms;
    if (launches) *launches = e.launches;
    return WFA_OK;
}

}  // extern "C"
