/*
 * wfa_hip.h -- C ABI of libwfa_hip.so, the MI355X (gfx950) backend for the per-record
 * waveform hot path of SnowingWolf/WaveformAnalysis.
 *
 * The reference is pure Python; there is no FFI to replace.  Each entry point below names
 * the reference function whose inner loop it replaces (paths relative to
 * waveform_analysis/).  The Python plugins in waveformanalysis_amd/plugins/ bind these
 * through ctypes (waveformanalysis_amd/_lib.py); INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C, caller-allocated host buffers, no callbacks;
 *   - every function returns 0 on success or a negative WFA_E_* code;
 *     wfa_last_error() returns the message of the calling thread's last failure;
 *   - one wfa_ctx per (GPU, host thread); a ctx owns one HIP stream and its device
 *     buffers; calls on one ctx are serialised by the caller;
 *   - the pool and the records SoA stay resident in HBM between calls, so
 *     filter -> hits -> features read them without another host transfer.
 */
#ifndef WFA_HIP_H
#define WFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFA_ABI_VERSION 1

#define WFA_OK 0
#define WFA_E_INVALID (-1)   /* bad argument / malformed records (maps to ValueError) */
#define WFA_E_HIP (-2)       /* HIP runtime failure */
#define WFA_E_STATE (-3)     /* call order: something required was not uploaded / planned */
#define WFA_E_NOMEM (-4)
#define WFA_E_RCCL (-5)
#define WFA_E_LIMIT (-6)     /* record longer than WFA_MAX_RECORD_SAMPLES etc. */

/* The plain threshold kernel keeps a record's hit bitmap in LDS (one 64-bit word per 64 samples, 4 waves per block,
 * 64 KiB): 2038 words.  The streaming hit kernel (k_sg_runs32) packs a run edge as (record << 16) | sample and is only
 * chosen for uniform records of at most 65 504 samples; longer uniform records take the bitmap route.  Everything else
 * indexes samples with 32-bit integers and sums them in 64 bits. */
#define WFA_MAX_RECORD_SAMPLES 130432
#define WFA_MAX_SG_WINDOW 63

/* wave source of a consumer (reference: cpu/_wave_source.py:119-165, records branch) */
#define WFA_SRC_RAW 0        /* wave_pool (uint16)                                        */
#define WFA_SRC_F32 1        /* wave_pool_filtered (float32) resident on the device       */
#define WFA_SRC_SG_FUSED 2   /* Savitzky-Golay of wave_pool evaluated on the fly, never   */
                             /* materialised (the fused baseline+filter+hitfind pass)     */

/* polarity code per record (reference: records["polarity"], dtypes.py:87) */
#define WFA_POL_UNKNOWN 0
#define WFA_POL_NEGATIVE 1
#define WFA_POL_POSITIVE 2
/* dense st_waveforms rule of BasicFeaturesPlugin (basic_features.py:243-262): wave-based formulas, positive sign;
 * every other stage treats it like WFA_POL_UNKNOWN */
#define WFA_POL_POSITIVE_WAVE 3

typedef struct wfa_ctx wfa_ctx;

int wfa_abi_version(void);
int wfa_device_count(int* count);
int wfa_last_error(char* buf, size_t buf_len);

int wfa_ctx_create(int device_id, wfa_ctx** out);
void wfa_ctx_destroy(wfa_ctx* ctx);
int wfa_sync(wfa_ctx* ctx);
/* Free the device scratch a later call rebuilds by itself (hit bitmap, run-event buffer, padded shadow pool, candidate
 * and sort buffers, filter scratch); the resident pools, the records, the plan and the rows of the last passes stay.
 * This is what Plugin.cleanup(context) of the HIP plugins calls (core/plugins/core/base.py:608-613: "releasing
 * resources" after compute()).  freed_bytes (nullable) receives the capacity given back. */
int wfa_release_scratch(wfa_ctx* ctx, int64_t* freed_bytes);
/* Choice between code paths that produce identical results (no counterpart in the reference: its plugins have one
 * code path).  For tests, which compare the paths with each other, and for measurement; a caller never needs it.
 * Names: "no_fast" (literal float64 hit kernel), "no_span" (per-record mask kernel), "no_pad" (no padded shadow
 * layout), "no_runs32" (bitmap route instead of the run-event kernel), "no_speculate" (exact row launches). */
int wfa_set_option(wfa_ctx* ctx, const char* name, int value);
/* GB/s of the last pool upload through the pinned staging ring (uploads of >= 4 MiB are copied chunk by chunk into two
 * pinned 32-MiB buffers while the previous chunk is on the wire; reference: the plugins hold host arrays only). */
int wfa_last_h2d_rate(wfa_ctx* ctx, double* gb_per_s);
/* Source of the device-resident THRESHOLD_HIT_DTYPE rows that wfa_hit_merge_count / wfa_group_hit_windows_count read
 * when ALL their column pointers are NULL: 1 = rows of the last hit pass (default), 2 = rows the last
 * wfa_rccl_gather_rows left on the root.  Replaces the host columns hit_merge.py:115-181 / event_grouping.py:418-471
 * take from the structured array: no upload, no download between the stages. */
int wfa_hit_rows_source(wfa_ctx* ctx, int which);

/* ---- resident inputs -------------------------------------------------------------------- */

/* wave_pool: flat uint16 samples (reference: processing/records_builder.py:197,299;
 * plugins/builtin/cpu/records.py:321-331). */
int wfa_upload_pool_u16(wfa_ctx* ctx, const uint16_t* pool, int64_t n_samples);

/* wave_pool_filtered produced elsewhere (reference: records.py:334-438 output).  With the same sample count as
 * the resident wave_pool it becomes its float32 twin; with a different count it replaces the pool set (the
 * uint16 pool is dropped and records must be uploaded again). */
int wfa_upload_pool_f32(wfa_ctx* ctx, const float* pool, int64_t n_samples);

/* Records index table as structure-of-arrays (reference row layout: processing/dtypes.py:80-100;
 * the per-channel option lookups of hit_finder.py:288-327 are resolved by the caller into
 * `threshold`).  Validates 0 <= offset, 0 <= length <= WFA_MAX_RECORD_SAMPLES,
 * offset+length <= pool size (reference: data/records_view.py:47-56). */
int wfa_upload_records_soa(wfa_ctx* ctx, int64_t n_records,
                           const int64_t* wave_offset, const int32_t* event_length,
                           const double* baseline, const int8_t* polarity_code,
                           const double* threshold, const int64_t* timestamp,
                           const int32_t* dt_ns, const int16_t* board, const int16_t* channel,
                           const int64_t* record_id);

/* The same table as packed rows, unpacked ON THE DEVICE: `rows` = n_records rows of row_bytes bytes exactly as a numpy
 * structured array holds them (reference RECORDS_DTYPE, core/processing/dtypes.py:80-100: 102 bytes, unaligned).
 * field_offsets[9] = byte offset inside a row of wave_offset (int64), event_length (int32), baseline (float64), polarity
 * (UCS-4 text of polarity_chars characters: "negative" / "positive" / anything else = unknown, as
 * data/records_view.py:87-100 reads it), timestamp (int64), dt (int32), board (int16), channel (int16), record_id (int64);
 * -1 for an absent polarity / dt / board / channel (defaults unknown / 1 / 0 / 0).  `threshold` applies to every record
 * unless `thresholds` (nullable, one float64 per record: hit_finder.py:288-327 resolved by the caller) is given;
 * `polarity` (nullable, one WFA_POL_* code per record) overrides the text field.  Same validation and messages as the
 * column route.  *record_ids_increasing = 0 tells the caller that record_id is not strictly increasing (it then has to
 * check uniqueness itself: records_view.py:383-400); *max_len = longest record. */
int wfa_upload_records_packed(wfa_ctx* ctx, const void* rows, int64_t n_records, int32_t row_bytes,
                              const int32_t* field_offsets, int32_t polarity_chars, double threshold,
                              const double* thresholds, const int8_t* polarity, int32_t* max_len,
                              int* record_ids_increasing);

/* Savitzky-Golay plan (reference: cpu/filtering.py:181-195,226-240 + scipy.signal.savgol_filter
 * mode="interp").  Tables are built on the host (waveformanalysis_amd/sg_plan.py):
 *   n_tables = (window+1)/2; table t serves effective window w = 2t+1 (short records,
 *   filtering.py:187-193); a table with w <= polyorder is unused (filter is a copy).
 *   tab: n_tables * (window + 2*(window/2)*window) doubles:
 *        [ fw[window] | E_left[window/2][window] | E_right[window/2][window] ] per table
 *        fw = correlation weights (savgol_coeffs reversed), E = polynomial edge projection.
 *   symmetric[t]: 1 if ndimage.correlate1d takes its symmetric branch for fw.
 *   Integer plan for the full window (exact rational arithmetic):
 *        itab: [ n[window] | NL[window/2][window] | NR[window/2][window] ] int32,
 *        y = n.x / den (interior), edges N.x / den_edge; guard_* = |numerator| below which
 *        the kernel evaluates the float64 chain instead.  int_ok = 0 disables it. */
int wfa_set_sg_plan(wfa_ctx* ctx, int window, int polyorder, const double* tab,
                    const uint8_t* symmetric, int int_ok, const int32_t* itab, int32_t den,
                    int32_t den_edge, int64_t guard, int64_t guard_edge);

/* ---- kernels ---------------------------------------------------------------------------- */

/* K1 baseline estimate: mean of samples [start, end) of every record as float64, NaN when
 * the window is empty (reference: records_builder.py:243-257, waveforms.py:714-733).
 * Writes the device-resident baseline column when update_records != 0; out may be NULL. */
int wfa_baseline_mean(wfa_ctx* ctx, int32_t start, int32_t end, int update_records, double* out);

/* Filter channel groups with different settings into ONE wave_pool_filtered (reference: per-channel
 * filter configs of build_filter_batches, cpu/filtering.py:339-374).  keep = 0 (default): wfa_savgol /
 * wfa_sosfiltfilt zero the whole output first (gaps stay 0.0, records.py:382) and then write the uploaded
 * records' slices.  keep = 1: the output is left as it is and only the uploaded records' slices are written. */
int wfa_filter_keep_output(wfa_ctx* ctx, int keep);

/* Copy the resident float32 pool (wave_pool_filtered as built or uploaded) to the host; n_samples must match. */
int wfa_download_pool_f32(wfa_ctx* ctx, float* out, int64_t n_samples);

/* K2 wave_pool_filtered: float32 pool aligned to wave_pool, gaps 0.0
 * (reference: records.py:368-438, filtering.py:377-407).  The result stays resident as the
 * WFA_SRC_F32 pool; out may be NULL. */
int wfa_savgol(wfa_ctx* ctx, float* out);

/* K3 wave_pool_filtered, Butterworth branch: scipy.signal.sosfiltfilt per record, float64 recursion,
 * float32 result (reference: filtering.py:84-101,198-224).  sos = n_sections x 6 coefficients from
 * scipy.signal.butter(..., output="sos"), zi = n_sections x 2 from scipy.signal.sosfilt_zi, padlen =
 * the reference's _estimate_sosfiltfilt_padlen; records with length <= padlen are copied.  The result
 * stays resident as the WFA_SRC_F32 pool; out may be NULL.  n_sections <= 8. */
int wfa_sosfiltfilt(wfa_ctx* ctx, int n_sections, const double* sos, const double* zi, int32_t padlen,
                    float* out);

/* K4 threshold hits (reference: hit_finder.py:231-255,329-413).  Two-phase so the caller can
 * allocate the structured array: _count runs the pass and returns the number of rows,
 * _fill copies them (THRESHOLD_HIT_DTYPE, 60-byte packed rows, order = record index, start).
 * max_len = padded matrix width of the reference call (max event_length over the whole
 * records array, hit_finder.py:364,370); pass 0 to use the maximum of the uploaded records. */
int wfa_threshold_hits_count(wfa_ctx* ctx, int source, int32_t left_extension,
                             int32_t right_extension, int32_t max_len, int64_t* n_hits);
int wfa_threshold_hits_fill(wfa_ctx* ctx, void* out_rows, int64_t n_hits);

/* K7 fused pass: baseline estimate over [bl_start, bl_end) (skipped if bl_end <= bl_start:
 * the uploaded baseline is used) + Savitzky-Golay + threshold hits in one read of the pool. */
int wfa_fused_baseline_filter_hits(wfa_ctx* ctx, int32_t bl_start, int32_t bl_end,
                                   int32_t left_extension, int32_t right_extension,
                                   int32_t max_len, int64_t* n_hits);

/* The same passes without a host round trip: _enqueue queues the whole pass on the context's stream and returns (the
 * row kernels are launched for the previous pass's row count + 12 % and read the real count on the device); _wait
 * blocks for the last enqueued pass and returns its row count, redoing the pass the exact way in the rare case that it
 * found more rows than that bound.  source = WFA_SRC_RAW / _F32 / _SG_FUSED; the baseline window applies to
 * WFA_SRC_SG_FUSED only (bl_end <= bl_start: uploaded baselines).  A context's first pass (no previous count) runs to
 * completion inside _enqueue.  wfa_threshold_hits_fill waits by itself.  (Streaming callers and bench.py use this
 * pair: consecutive chunks' passes run back to back on the device.) */
int wfa_hits_enqueue(wfa_ctx* ctx, int source, int32_t bl_start, int32_t bl_end, int32_t left_extension,
                     int32_t right_extension, int32_t max_len);
int wfa_hits_wait(wfa_ctx* ctx, int64_t* n_hits);

/* K8 find_peaks-based hit detector, records source (reference: cpu/peak_finding.py:395-614 calling
 * scipy.signal.find_peaks(det, height, distance, prominence, width, threshold) with scalar lower bounds, then
 * _calculate_peak_height 567-614).  source: WFA_SRC_RAW or WFA_SRC_F32.  Two-phase like the threshold hits; rows
 * are HIT_DTYPE (48 B packed: position i8, height f4, integral f4, edge_start f4, edge_end f4, dt i4,
 * timestamp i8, board i2, channel i2, record_id i8) in (record, position) order.  has_threshold = 0 means
 * threshold=None.  An empty minmax window fails with numpy's "zero-size array ..." message (WFA_E_INVALID). */
#define WFA_HEIGHT_MINMAX 0
#define WFA_HEIGHT_DIFF 1
/* signal_mode: WFA_PEAK_SIGNAL_RECORDS = records branch (signal = -rv.signals(id): float32 (w - baseline), sign by
 * polarity; detection on diff(signal) or signal).  WFA_PEAK_SIGNAL_ROWS = dense st_waveforms / filtered_waveforms
 * branch (peak_finding.py:316-378): the stored samples are the waveform, pulses are negative-going (detection on
 * -diff(row) in the row's dtype or float64 baseline - row), the height is measured on the row itself. */
#define WFA_PEAK_SIGNAL_RECORDS 0
#define WFA_PEAK_SIGNAL_ROWS 1
/* WFA_PEAK_SIGNAL_ROWS_F64 = the streaming detector (streaming/cpu/signal_peaks.py:226-406): like _ROWS, but the row
 * is converted to float64 first (detection on -diff in float64), rows are never cut at event_length, and the 'diff'
 * height is cumsum(-diff(row))[end] - cumsum[start] with numpy's sequential cumsum, rounded to float32. */
#define WFA_PEAK_SIGNAL_ROWS_F64 2
int wfa_find_peaks_count(wfa_ctx* ctx, int source, int signal_mode, int use_derivative, double height, int has_threshold,
                         double threshold, int32_t distance, double prominence, double width, int height_method,
                         int32_t height_window_extension, int64_t* n_peaks);
int wfa_find_peaks_fill(wfa_ctx* ctx, void* out_rows, int64_t n_peaks);

/* K5 basic features (reference: basic_features.py:108-195).  Ranges are python slice bounds;
 * *_has_end = 0 means "None".  fixed_baseline: per-record override, NaN = none, may be NULL.
 * out: BASIC_FEATURES_DTYPE rows (36 B). */
int wfa_basic_features(wfa_ctx* ctx, int source, int64_t height_start, int64_t height_end,
                       int height_has_end, int64_t area_start, int64_t area_end, int area_has_end,
                       const double* fixed_baseline, void* out_rows);

/* K14 legacy threshold crossings on dense rows (reference: processing/event_grouping.py:46-95 `find_hits`):
 * mask = (baselines[row] - wave) > threshold in float64, one hit per 0 -> 1 transition.  The resident pool is the
 * row-major wave matrix (as for K10).  fill: event_index / start sample of every hit in row-major order. */
int wfa_find_hits_count(wfa_ctx* ctx, int source, int64_t n_rows, int32_t row_length, const double* baselines,
                        double threshold, int64_t* n_hits);
int wfa_find_hits_fill(wfa_ctx* ctx, int64_t n_hits, int64_t* event_index, int64_t* start_sample);

/* K10 rise/fall/total width per hit on dense waveform rows (reference: cpu/waveform_width.py:205-374).
 * The resident pool is the row-major (n_rows x row_length) wave matrix of st_waveforms (source WFA_SRC_RAW,
 * non-negative int16 ADC codes viewed as uint16) or filtered_waveforms (WFA_SRC_F32).  Per hit: position and
 * the row it refers to (row_index < 0: no such row, hit skipped).  out: WAVEFORM_WIDTH_DTYPE rows (56 B) with the
 * six width fields, peak_position and peak_height filled, the id fields zero (the caller copies them from the
 * hit table); valid[i] = 0 where the reference drops the hit (position past the row, peak value <= 0).
 * Arithmetic follows numpy's promotion rules for the source dtype (float64 for int16 rows, float32 for
 * float32 rows incl. its mixed python-float cases), so the float fields are bit-exact. */
int wfa_waveform_width(wfa_ctx* ctx, int source, int64_t n_hits, const int64_t* position,
                       const int64_t* row_index, int64_t n_rows, int32_t row_length, double rise_low,
                       double rise_high, double fall_high, double fall_low, double sampling_rate,
                       int interpolation, void* out_rows, uint8_t* valid);

/* ---- hit-table stages (device sort + scans; host columns in, host index tables out) ----------------------
 * K11 hit merge (reference: cpu/hit_merge.py:115-181 `_build_merged_clusters`, 256-322 `_emit_cluster`).
 * Input: the columns of a hit table (THRESHOLD_HIT_DTYPE or its renamed variants).
 * count/fill = HitMergeClustersPlugin: per hardware channel (ascending board, channel) hits are ordered by
 * abs_start = ts + (edge_start - position) * dt * 1e3 (float64, stable) and chained while merge_gap_ns > 0, dt
 * unchanged, gap <= merge_gap and total width <= max_total_width.  fill: order[n] = hit index of each
 * cluster-ordered row (= hit_merge_clusters.hit_index) and cluster_offset[n_clusters + 1] into it.
 * emit = the per-cluster part of HitMergePlugin for ANY membership table (as the reference derives hit_merged
 * from whatever hit_merge_clusters holds): anchor hit index (max height, tie -> smallest timestamp, first such),
 * max height, sum of integrals (numpy's pairwise float64 order, as float32), min/max sample window (-1, -1 if the
 * cluster spans records) and width (-1 likewise). */
int wfa_hit_merge_count(wfa_ctx* ctx, int64_t n_hits, const int64_t* timestamp, const int64_t* position,
                        const int32_t* edge_start, const int32_t* edge_end, const int32_t* dt, const int16_t* board,
                        const int16_t* channel, double merge_gap_ns, double max_total_width_ns, int64_t* n_clusters);
int wfa_hit_merge_fill(wfa_ctx* ctx, int64_t n_hits, int64_t n_clusters, int64_t* order, int64_t* cluster_offset);
int wfa_hit_merge_emit(wfa_ctx* ctx, int64_t n_hits, const int64_t* timestamp, const int32_t* sample_start,
                       const int32_t* sample_end, const int64_t* record_id, const float* height, const float* integral,
                       int64_t n_members, const int64_t* member_hit, int64_t n_clusters, const int64_t* cluster_offset,
                       int64_t* anchor, float* out_height, float* out_integral, int32_t* out_start, int32_t* out_end,
                       float* out_width);

/* K9 event grouping (reference: processing/event_grouping.py:286-471 `group_hit_windows`).  Absolute windows as
 * above from (sample_start, sample_end); global order np.lexsort((record_id, timestamp, dt, abs_start)); a hit
 * opens a new event when abs_start > running max(abs_end) + time_window_ns * 1e3; inside an event hits are ordered
 * by (board, channel, dt, abs_start, timestamp, record_id).  abs_*_fix (both or neither, may be NULL): where not
 * NaN they replace the computed window (merged hits spanning records take the extent of their components,
 * event_grouping.py:371-416).  fill: order[n] (hit indices, event-major),
 * event_start[n_events + 1], t_min / t_max = int(min abs_start) / int(max abs_end) per event (ps). */
int wfa_group_hit_windows_count(wfa_ctx* ctx, int64_t n_hits, const int64_t* timestamp, const int64_t* position,
                                const int32_t* sample_start, const int32_t* sample_end, const int32_t* dt,
                                const int16_t* board, const int16_t* channel, const int64_t* record_id,
                                const double* abs_start_fix, const double* abs_end_fix, double time_window_ns,
                                int64_t* n_events);
int wfa_group_hit_windows_fill(wfa_ctx* ctx, int64_t n_hits, int64_t n_events, int64_t* order, int64_t* event_start,
                               int64_t* t_min, int64_t* t_max);

/* Legacy fixed-window grouping: group_multi_channel_hits (processing/event_grouping.py:98-283) with the boundary search of
 * _find_cluster_boundaries_numba (475-525).  The hits are ordered by timestamp (stable; pandas' default sort leaves the
 * order of equal timestamps unspecified), a cluster takes every hit whose timestamp is <= (double)first + time_window_ps
 * (float64 comparison, as numpy / numba compare an int64 column with a float64 needle), and inside a cluster the hits
 * stand by channel (stable).  order[k] = input row of output row k (event-major); bounds[e] .. bounds[e + 1] = the rows of
 * event e in `order`.  time_window_ps = time_window_ns * 1e3, formed by the caller as the reference forms it. */
int wfa_group_multi_channel_count(wfa_ctx* ctx, int64_t n_hits, const int64_t* timestamp, const int64_t* channel,
                                  double time_window_ps, int64_t* n_events);
int wfa_group_multi_channel_fill(wfa_ctx* ctx, int64_t n_hits, int64_t n_events, int64_t* order, int64_t* bounds);

/* ---- records builder (reference: processing/records_builder.py) -------------------------------------------------
 * K12 global record order: np.lexsort((seq, channel, board, pid, timestamp)) (records_builder.py:115-120), i.e. the
 * order a k-way merge of sorted parts produces (341-426, 869-945).  order[k] = source row of output row k. */
int wfa_records_sort(wfa_ctx* ctx, int64_t n_records, const int64_t* timestamp, const int32_t* pid,
                     const int16_t* board, const int16_t* channel, int64_t* order);

/* K13 pack the wave slices of the records, given in OUTPUT order, into one contiguous pool
 * (records_builder.py:195-207, 400-409).  src_pool: the concatenated source samples (uint16 bit patterns);
 * out_offset[r] receives the running sum of max(length, 0).  The packed pool becomes the resident wave_pool of the
 * context (records must be uploaded again); out_pool, when not NULL, also receives it.  src_pool = NULL with
 * src_samples > 0 takes the samples the last wfa_csv_decode_fill left on the device. */
int wfa_pool_gather(wfa_ctx* ctx, int64_t n_records, const int64_t* src_offset, const int32_t* length,
                    const uint16_t* src_pool, int64_t src_samples, int64_t* out_offset, uint16_t* out_pool,
                    int64_t out_samples);

/* K15 CAEN VX2730 CSV text -> integers on the device (reference: utils/formats/vx2730.py:78-110 column layout,
 * :193-340 `VX2730Reader.read_file` -- its polars / pyarrow / pandas backends all yield these integers; consumer
 * processing/records_builder.py:212-302).  text = the bytes of one or more files after their header rows: rows end
 * with '\n' (a '\r' before it is dropped, a last row without '\n' counts), fields are separated by `delimiter`.
 * _count uploads the text, indexes the rows and counts their fields; n_rows includes empty rows (0 fields).
 * n_samples = sum over rows of max(n_fields - samples_start, 0).
 * _fill parses columns meta_cols[0..n_meta) (each < samples_start) as int64 into meta[n_rows x n_meta] and the fields
 * from samples_start on as uint16 ADC codes into the ragged pool samples[sample_offset[r] ...]; row_offset[r] = byte
 * offset of row r in text (maps rows back to files), n_fields[r] = its field count.  Any output pointer but meta may
 * be NULL.  The samples stay resident: wfa_pool_gather with src_pool = NULL, src_samples = n_samples packs them
 * without a host round trip.  A requested field that is not a decimal integer, or a sample outside 0..65535, fails
 * with WFA_E_INVALID naming the row and field; other columns (ENERGY, FLAGS as hex, ...) are never parsed. */
int wfa_csv_decode_count(wfa_ctx* ctx, const uint8_t* text, int64_t n_bytes, int delimiter, int32_t samples_start,
                         int64_t* n_rows, int64_t* n_samples);
int wfa_csv_decode_fill(wfa_ctx* ctx, int64_t n_rows, int32_t n_meta, const int32_t* meta_cols, int64_t* meta,
                        int64_t* row_offset, int32_t* n_fields, int64_t* sample_offset, uint16_t* samples,
                        int64_t n_samples);

/* Wave index of a CAEN V1725 DAW_DEMO binary stream held in host memory (reference: utils/formats/v1725.py:66-114
 * `V1725Reader.iter_waves`): 16-byte event header (channel mask in bytes 4 and 11), per set channel a 12-byte
 * header (size in 32-bit words: 22 bits, truncation flag bit 6 of byte 3, 48-bit timestamp bytes 4-9, uint16 baseline
 * bytes 10-11) + int16 payload.  Host-only, no context.  Call with capacity 0 to count, then with arrays of n_waves
 * entries.  payload_offset is in bytes from buf; a short header or payload ends the stream like the reference's reader. */
int wfa_v1725_index(const uint8_t* buf, int64_t n_bytes, int64_t capacity, int16_t* channel, int64_t* timestamp,
                    uint8_t* trunc, uint16_t* baseline, int64_t* payload_offset, int32_t* n_samples, int64_t* n_waves);

/* K6 integral-quantile width (reference: waveform_width_integral.py:166-227).
 * out: WAVEFORM_WIDTH_INTEGRAL_DTYPE rows (52 B). */
int wfa_width_integral(wfa_ctx* ctx, int source, double q_low, double q_high, double dt,
                       void* out_rows);

/* BasicFeaturesPlugin and WaveformWidthIntegralPlugin records branches (cpu/basic_features.py:108-195,
 * cpu/waveform_width_integral.py:83-231) on the raw wave_pool in ONE read of the pool: BASELINE config 3 asks for both
 * tables of the same records.  Uniform records, area range = whole record: one kernel stages every group of records once
 * and forms both rows (the same additions in the same order as the two separate kernels: identical bytes); any other
 * layout / range runs the two kernels one after the other.  out_basic (36-byte rows) / out_width (52-byte rows) may be NULL:
 * the tables then stay on the device only. */
int wfa_features_both(wfa_ctx* ctx, int64_t height_start, int64_t height_end, int height_has_end, int64_t area_start,
                      int64_t area_end, int area_has_end, double q_low, double q_high, double dt, void* out_basic,
                      void* out_width);

/* ---- measurement ------------------------------------------------------------------------ */

/* on = 1: every kernel launch is bracketed by HIP events on the ctx stream.  on = 2: only the streaming (mask) kernel
 * of the fused / SG-fused hit pass is -- the pair of events around each of its small follow-up kernels costs about as
 * much as the gap it measures.  on = 0: off. */
int wfa_profile_enable(wfa_ctx* ctx, int on);
int wfa_profile_reset(wfa_ctx* ctx);
/* idx-th profiled kernel: name (<= name_len), summed milliseconds, launches. Returns
 * WFA_E_INVALID past the end. */
int wfa_profile_get(wfa_ctx* ctx, int idx, char* name, size_t name_len, double* total_ms,
                    int64_t* launches);

/* ---- multi-GPU gather for event grouping (RCCL over xGMI) -------------------------------- */

/* 128-byte unique id created on rank 0 and broadcast by the launcher. */
int wfa_rccl_unique_id(void* id128);
int wfa_rccl_init(wfa_ctx* ctx, int rank, int n_ranks, const void* id128);
/* Step 1: every rank learns counts[n_ranks] (one int64 all-gather). */
int wfa_rccl_allgather_counts(wfa_ctx* ctx, int64_t n_rows, int64_t* counts);
/* Step 2: all ranks send n_rows rows of row_bytes each to `root` (grouped send/recv); the
 * root receives them concatenated in rank order into out (host buffer of sum(counts) rows,
 * ignored elsewhere).  rows == NULL sends the device-resident rows of the last hit pass
 * (row_bytes must be 60) without a host round trip.  counts = result of step 1. */
int wfa_rccl_gather_rows(wfa_ctx* ctx, const void* rows, int64_t n_rows, int32_t row_bytes,
                         int root, const int64_t* counts, void* out);
/* on != 0: the 60-byte rows of the following exchanges are appended on the root behind the rows already gathered (a rank
 * that works through several shards or time-range chunks sends one exchange per chunk; event grouping,
 * event_grouping.py:286-471, then reads one table: wfa_hit_rows_source(ctx, 2)).  `out` of such an exchange receives
 * only that exchange's rows.  on == 0: back to one table per exchange; the gathered table is dropped. */
int wfa_rccl_gather_append(wfa_ctx* ctx, int on);
int wfa_rccl_destroy(wfa_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* WFA_HIP_H */
