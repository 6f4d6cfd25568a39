// Kernel parameter blocks and launchers (implemented in wfa_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "wfa_hip.h"

namespace wfa {

struct PoolView {
    const uint16_t* u16;
    const float* f32;
    int64_t n;
};

// Records SoA (reference row layout: core/processing/dtypes.py:80-100).
struct RecView {
    int64_t R;
    const int64_t* off;
    const int32_t* len;
    const double* baseline;
    double* baseline_rw;  // same column, writable (fused baseline estimate)
    const int8_t* pol;
    const double* thr;
    const int64_t* ts;
    const int32_t* dt;
    const int16_t* board;
    const int16_t* chan;
    const int64_t* rid;
    const int64_t* bm_off;  // byte offset of the record's mask bits in the hit bitmap
};

struct SgParams {
    int W, P, H;
    int stride;  // doubles per table
    const double* tab;
    const uint8_t* sym;
    int int_ok;
    const int32_t* itab;
    int32_t den, den_edge;
    int64_t guard, guard_edge;
    double rden, rden_edge;
    int32_t margin;       // numerator units covering one float32 ulp of y (candidate band half-width)
    int32_t margin_edge;  // same for the edge projection rows (den_edge)
};

struct HitParams {
    int32_t le, re, max_len;
    int32_t bl_start, bl_end;
    int32_t bm_words;  // LDS bitmap words per wave
    int32_t chunk_rows;
    int32_t use_fast;  // allow the integer fast path (0 = literal float64 kernel, for A/B tests)
    uint8_t* tmp;
    int64_t tmp_rows;
    unsigned long long* cursor;
    int64_t* rec_tmp_start;
    int32_t* rec_nhits;
};

// fused mask pass (k_sg_mask)
struct MaskParams {
    int32_t bl_start, bl_end;
    uint8_t* bitmap;
    int32_t* rec_nhits;
};

// span mode of the mask pass: all records have length L (multiple of 8, >= 24), are contiguous from
// pool index off0 (multiple of 8) and share one polarity class
struct SpanParams {
    int64_t off0;
    int32_t L;
    int32_t positive;
    int32_t rs;        // records per span (<= 64)
    int64_t n_spans;
    int64_t bm_off0;   // bitmap byte offset of record 0
    int64_t bm_stride; // bitmap bytes per record
    int32_t S = 0;     // record stride in samples (0 = L: packed records; span16 / savgol_span: multiple of 16, S - L < 16)
    int64_t out_off0 = 0;  // k_savgol_span on the padded layout: pool position of record 0 in the packed output
};

// run events of the streaming pass (k_sg_runs32): per span a contiguous block of 32-bit events
// (record-in-span << 16 | position), start / end alternating, allocated from `cursor` when the span is done
struct RunsParams {
    uint32_t* ev;
    int64_t ev_cap;              // events the buffer holds
    unsigned long long* cursor;  // events allocated so far (keeps counting beyond ev_cap: the size a redo needs)
    int64_t* span_off;           // [n_spans] first event of the span
    int32_t* span_cnt;           // [n_spans] hits (= events / 2) of the span
    int32_t* flags;              // bit 0: a span outgrew the LDS buffer, bit 1: `ev` too small
    unsigned long long* group_sum;  // [ceil(n_spans / 64)] hits of 64 consecutive spans (atomic adds of the spans' flushes)
    uint32_t* lit_cnt;              // RowParams::lit_cnt (k_runs_to_desc clears it)
};

// k_sg_runs32: arguments the tile loop keeps in scalar registers, and (by pointer, device-resident) what only the
// float64 reference paths read
struct RunsCold {
    PoolView pool;
    SgParams sg;
};
struct RunsArgs {
    const uint16_t* pool;   // uint16 pool the records live in (packed, or the padded shadow)
    const double* thr;      // records.threshold
    double* baseline;       // records.baseline (read when given, written when estimated in-stream)
    int64_t R;
    const int32_t* itab;    // integer SG plan: W numerators, then the 2H edge rows
    int32_t den, margin;
    int32_t den_edge, margin_edge;  // edge projection rows: denominator, band half-width
    double delta;           // numerator units by which scipy's float64 chain may differ from the exact rational
    int32_t W, L, S, positive, rs;
    int32_t wstride, nseg, segw;  // sg_runs32_geometry: LDS words per record (odd), flush segments per record, words per segment
    int32_t dep;            // sg_runs32_deposit: the records' last lanes go to LDS in the tile loop
    int64_t off0, n_spans;
    uint32_t* ev;
    int64_t ev_cap;
    unsigned long long* cursor;
    int64_t* span_off;
    int32_t* span_cnt;
    int32_t* flags;
    unsigned long long* group_sum;  // see RunsParams
    const RunsCold* cold;
};

// hit-row pass (k_hit_rows)
struct RowParams {
    int32_t le, re, max_len;
    int32_t fast_halo;  // SG half window when the integer row kernel may be used, else 0
    int32_t stage_bytes = 48 * 1024;  // LDS staging of k_hit_runs: mask bytes of 256 consecutive records (0 = none)
    // speculative launch (no host round trip for the hit count): the kernels are launched for `cap` rows, the row
    // count of this pass is read from `n_dev` on the device, nothing is written at or beyond row `cap`
    int64_t cap = 0;
    const int64_t* n_dev = nullptr;
    // uniform records (span mode): wave_offset = uni_off0 + r * uni_L and one polarity for all, so the row kernel
    // needs no per-record loads before its first sample chunk (uni_L = 0: read them from the records table)
    int32_t uni_L = 0, uni_positive = 0, uni_S = 0;  // uni_S: stride in samples (0 = uni_L)
    int64_t uni_off0 = 0;
    // end of a queued streaming pass (k_hit_rows_literal is its last kernel): one thread copies the pass's row count and the
    // two control words (event cursor, overflow flags) to `pass_report` (pinned host memory) and clears the control words
    // for the next pass -- what two device-to-host copies and a memset did as three more launches per pass
    int64_t* pass_report = nullptr;
    unsigned long long* pass_ctrl = nullptr;
    unsigned long long* pass_groups = nullptr;  // the pass's group sums (RunsParams::group_sum), cleared as well
    int64_t pass_n_groups = 0;
    // hits the fast row kernel hands to the literal kernel, as a list (count cleared by k_runs_to_desc at the start of the
    // pass): the literal kernel then reads the list instead of every descriptor's flag; more than lit_cap entries -> the
    // flags decide as before
    uint32_t* lit_cnt = nullptr;
    int32_t* lit_list = nullptr;
    int32_t lit_cap = 0;
};

// find_peaks-based hit detector (k_find_peaks): scalar lower bounds only, as the reference plugin passes them
struct PeakParams {
    int use_derivative;
    double hmin;
    int has_threshold;
    double tmin;
    int distance;
    double pmin;
    double wmin;
    int ext;
    int height_diff;  // 1: height_method 'diff', 0: 'minmax'
    int rows;         // WFA_PEAK_SIGNAL_* (0 records branch, 1 dense rows, 2 dense rows as float64)
};

struct FeatParams {
    int64_t h0, h1, a0, a1;
    int h_has_end, a_has_end;
    const double* fixed_bl;  // nullable
};

struct WidthParams {
    double q_low, q_high, dt;
};

hipError_t launch_baseline_mean(hipStream_t st, const PoolView& pool, const RecView& rec,
                                int32_t start, int32_t end, double* out);
hipError_t launch_savgol(hipStream_t st, const PoolView& pool, const RecView& rec,
                         const SgParams& sg, float* out);
int hits_grid(int64_t R);
int hits_waves(int64_t R);
hipError_t launch_hits(hipStream_t st, int source, bool fused_baseline, const PoolView& pool,
                       const RecView& rec, const SgParams& sg, const HitParams& hp);
int64_t scan_blocks_for(int64_t n);
hipError_t launch_scan(hipStream_t st, const int32_t* counts, int64_t n, int64_t* block_sums,
                       int64_t* out);
hipError_t launch_hits_gather(hipStream_t st, const uint8_t* tmp, const int64_t* tmp_start,
                              const int32_t* nhits, const int64_t* out_start, int64_t R, uint8_t* out);
hipError_t launch_sosfiltfilt(hipStream_t st, const PoolView& pool, const RecView& rec, int n_sections,
                              const double* sos, const double* zi, int edge, int64_t r_begin, int64_t r_end,
                              double* scratch, int64_t batch_stride, float* out);
hipError_t launch_find_hits_legacy(hipStream_t st, int source, bool fill, const PoolView& pool, int64_t n_rows, int32_t L,
                                   const double* baselines, double threshold, int32_t* counts, const int64_t* out_start,
                                   int64_t* out_event, int64_t* out_time);
hipError_t launch_waveform_width(hipStream_t st, int source, const PoolView& pool, int64_t n_hits,
                                 const int64_t* position, const int64_t* row_index, int64_t n_rows, int32_t L,
                                 double rise_low, double rise_high, double fall_high, double fall_low,
                                 double sampling_rate, int interpolation, uint8_t* out, uint8_t* valid);
// find_peaks: candidate scan per record (count / fill), distance selection per record, then per candidate
hipError_t launch_find_peaks(hipStream_t st, int source, bool fill, const PoolView& pool, const RecView& rec,
                             const PeakParams& pp, int32_t* counts, const int64_t* out_start, int32_t* cand_pos,
                             double* cand_val, int64_t* cand_rec);
constexpr int kPeakSlots = 8;  // candidates per record the single walk keeps (more: the fill walk runs)
hipError_t launch_find_peaks_slots(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                                   int K, int32_t* counts, int32_t* slot_pos, double* slot_val, int* overflow);
bool launch_find_peaks_hot(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                           int64_t off0, int L, int K, int32_t* counts, int32_t* slot_pos, double* slot_val, int* overflow,
                           hipError_t* err);
bool launch_find_peaks_staged(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                              int64_t off0, int L, int K, int32_t* counts, int32_t* slot_pos, double* slot_val, int* overflow,
                              hipError_t* err);
hipError_t launch_peak_compact(hipStream_t st, int64_t R, int K, const int32_t* counts, const int64_t* cand_start,
                               const int32_t* slot_pos, const double* slot_val, int32_t* cand_pos, double* cand_val,
                               int64_t* cand_rec);
hipError_t launch_peak_select(hipStream_t st, int64_t R, const int32_t* counts, const int64_t* cand_start,
                              const int32_t* cand_pos, const double* cand_val, uint8_t* state, int distance);
hipError_t launch_peak_eval(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                            int64_t n_cand, const int64_t* cand_rec, const int32_t* cand_pos, const uint8_t* state,
                            int32_t* accept, double* ips);
hipError_t launch_peak_rows(hipStream_t st, int source, const PoolView& pool, const RecView& rec, const PeakParams& pp,
                            int64_t n_cand, const int64_t* cand_rec, const int32_t* cand_pos, const int32_t* accept,
                            const int64_t* row_start, const double* ips, uint8_t* out, int* err);
int hit_runs_block();  // records per block of k_hit_runs (sizes its LDS staging)
bool sg_mask_supported(const SgParams& sg);
bool sg_mask_span16_padded_supported(const SgParams& sg, int32_t L);
hipError_t launch_pad_rows(hipStream_t st, const uint16_t* src, int64_t off0, int32_t L, int32_t S, int64_t R,
                           uint16_t* dst, int64_t* dst_off);
hipError_t launch_sg_mask(hipStream_t st, bool fused_baseline, int max_len, const PoolView& pool,
                          const RecView& rec, const SgParams& sg, const MaskParams& mp);
hipError_t launch_savgol_span(hipStream_t st, const PoolView& pool, const RecView& rec, const SgParams& sg,
                              const SpanParams& sp, float* out);
bool sg_mask_span16_supported(const SgParams& sg, int L);
hipError_t launch_sg_mask_span16(hipStream_t st, bool fused_baseline, const PoolView& pool, const RecView& rec,
                                 const SgParams& sg, const MaskParams& mp, const SpanParams& sp);
bool sg_runs32_supported(const SgParams& sg, int32_t L, int32_t S, int32_t bl_start, int32_t bl_end, bool fused_bl);
int64_t sg_runs32_event_slot();  // events of the buffer every span owns
void sg_runs32_geometry(int32_t S, int32_t* rs, int32_t* wstride, int32_t* nseg, int32_t* segw);
int64_t sg_runs32_lds_words(int32_t rs, int32_t wstride, bool dep);
bool sg_runs32_deposit(int32_t L, int32_t S, int32_t W, int32_t rs, int32_t wstride);
hipError_t launch_sg_runs32(hipStream_t st, bool fused_baseline, const RunsArgs& a);
hipError_t launch_runs_to_desc(hipStream_t st, const RunsParams& rp, int64_t n_spans, int32_t rs,
                               int64_t* total_out, int64_t cap, int4* desc);
hipError_t launch_hit_runs(hipStream_t st, const RecView& rec, const uint8_t* bitmap, const int32_t* nhits,
                           const int64_t* out_start, int4* desc, const RowParams& rp);
// grouped: the 8-lanes-per-hit kernel of rounds 1-2 (wfa_set_option "rows_grouped"); default: the flat chunk-per-lane kernel
hipError_t launch_hit_rows_fast(hipStream_t st, const PoolView& pool, const RecView& rec, const SgParams& sg,
                                const RowParams& rp, int4* desc, int64_t n_hits, uint8_t* out, bool grouped = false);
hipError_t launch_hit_rows_literal(hipStream_t st, int source, const PoolView& pool, const RecView& rec,
                                   const SgParams& sg, const RowParams& rp, const int4* desc, int64_t n_hits,
                                   bool only_flagged, uint8_t* out);
struct FeatParams;
struct WidthParams;
}  // namespace wfa
struct wfa_ctx;
namespace wfa {
// uniform records: wave-per-record kernels (wfa_features.hip); false = layout not covered, launch the general kernel
bool launch_basic_features_wave(wfa_ctx* c, const RecView& rec, const FeatParams& fp, uint8_t* out, hipError_t* err);
bool launch_width_integral_wave(wfa_ctx* c, const RecView& rec, const WidthParams& wp, uint8_t* out, hipError_t* err);
bool launch_features_both_wave(wfa_ctx* c, const RecView& rec, const FeatParams& fp, const WidthParams& wp, uint8_t* out_basic,
                               uint8_t* out_width, hipError_t* err);
hipError_t launch_basic_features(hipStream_t st, int source, const PoolView& pool, const RecView& rec,
                                 const SgParams& sg, const FeatParams& fp, uint8_t* out);
hipError_t launch_width_integral(hipStream_t st, int source, const PoolView& pool, const RecView& rec,
                                 const SgParams& sg, const WidthParams& wp, uint8_t* out);

}  // namespace wfa
