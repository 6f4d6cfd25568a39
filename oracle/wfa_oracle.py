"""CPU oracle for the per-record waveform hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy restatement of the reference algorithms
(SnowingWolf/WaveformAnalysis, files under waveform_analysis/).  It is the
*checker*: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it.  The product (waveformanalysis_amd/) never imports it and has no
CPU fallback.

Parity status: PINNED.  tests/golden/*.npz hold inputs and outputs produced by
running the reference's own plugins in the build container
(tests/golden/make_golden.py); tests/test_oracle_golden.py checks every function
below against them bit-for-bit (floats included).

Third-party arithmetic: like the reference, the Savitzky-Golay / Butterworth
filters are scipy's (`scipy.signal.savgol_filter`, `butter`, `sosfiltfilt`;
reference call sites core/plugins/builtin/cpu/filtering.py:19,101,223,234;
scipy is pinned only as >=1.7.0 in the reference's pyproject.toml:31, the image
has 1.15.3).

Each function cites the reference lines it follows.  "literal" functions keep
the reference's per-record / per-hit loops (they are also what bench.py times as
the CPU baseline); "*_uniform" helpers are vectorised equivalents for runs with
equal-length contiguous records, used for parity at 10^6-10^8 samples and checked
against the literal forms in tests/test_oracle_golden.py.
"""

from __future__ import annotations

import math

import numpy as np
from scipy.signal import butter, savgol_filter, sosfiltfilt

# --- layouts (reference: core/processing/dtypes.py:80-100, hit_finder.py:33-49,
#     basic_features.py:29-40, waveform_width_integral.py:25-39) ------------------------------
RECORDS_DTYPE = np.dtype(
    [
        ("timestamp", "i8"), ("pid", "i4"), ("board", "i2"), ("channel", "i2"),
        ("baseline", "f8"), ("baseline_upstream", "f8"), ("polarity", "U8"),
        ("record_id", "i8"), ("dt", "i4"), ("trigger_type", "i2"), ("flags", "u4"),
        ("wave_offset", "i8"), ("event_length", "i4"), ("time", "i8"),
    ]
)
THRESHOLD_HIT_DTYPE = np.dtype(
    [
        ("position", "i8"), ("height", "f4"), ("integral", "f4"), ("edge_start", "i4"),
        ("edge_end", "i4"), ("width", "f4"), ("dt", "i4"), ("rise_time", "f4"),
        ("fall_time", "f4"), ("timestamp", "i8"), ("board", "i2"), ("channel", "i2"),
        ("record_id", "i8"),
    ]
)
BASIC_FEATURES_DTYPE = np.dtype(
    [
        ("height", "f4"), ("amp", "f4"), ("area", "f4"), ("max_abs_diff", "f4"),
        ("timestamp", "i8"), ("board", "i2"), ("channel", "i2"), ("event_index", "i8"),
    ]
)
WAVEFORM_WIDTH_INTEGRAL_DTYPE = np.dtype(
    [
        ("t_low", "f4"), ("t_high", "f4"), ("width", "f4"), ("t_low_samples", "f4"),
        ("t_high_samples", "f4"), ("width_samples", "f4"), ("q_total", "f8"),
        ("timestamp", "i8"), ("board", "i2"), ("channel", "i2"), ("event_index", "i8"),
    ]
)


# --- a3: baseline estimate --------------------------------------------------------------------
def baseline_mean(raw: np.ndarray, start: int, end: int) -> np.ndarray:
    """records_builder.py:243-257: mean over raw[:, start:end] as float64, NaN if empty."""
    raw = np.asarray(raw)
    end = min(int(end), raw.shape[1])
    if end <= start:
        return np.full(len(raw), np.nan, dtype=np.float64)
    return np.mean(raw[:, start:end].astype(float), axis=1)


# --- a4: filters ------------------------------------------------------------------------------
def sg_window_length(n_samples: int, sg_window_size: int, sg_poly_order: int):
    """filtering.py:181-195: effective odd window, or None when the filter is a no-op."""
    window = min(int(sg_window_size), int(n_samples))
    if window % 2 == 0:
        window -= 1
    if window <= int(sg_poly_order):
        return None
    return window


def sosfiltfilt_padlen(sos: np.ndarray) -> int:
    """filtering.py:198-203 (scipy's default pad-length heuristic)."""
    n_sections = int(sos.shape[0])
    zeros_at_origin = int((sos[:, 2] == 0).sum())
    poles_at_origin = int((sos[:, 5] == 0).sum())
    return 3 * (2 * n_sections + 1 - min(zeros_at_origin, poles_at_origin))


def design_bw(lowcut=0.1, highcut=0.5, fs=0.5, order=4) -> np.ndarray:
    """filtering.py:84-101."""
    return butter(int(order), [float(lowcut), float(highcut)], btype="band", output="sos", fs=float(fs))


def apply_filter_core(waves_f32: np.ndarray, filter_type: str = "SG", *, bw_sos=None,
                      sg_window_size: int = 11, sg_poly_order: int = 2) -> np.ndarray:
    """filtering.py:206-241 for a 1-D wave or a 2-D batch (axis=-1)."""
    waves_f32 = np.asarray(waves_f32, dtype=np.float32)
    if filter_type == "BW":
        if waves_f32.shape[-1] <= sosfiltfilt_padlen(bw_sos):
            return np.array(waves_f32, copy=True)
        return np.asarray(sosfiltfilt(bw_sos, waves_f32, axis=-1), dtype=np.float32)
    window = sg_window_length(waves_f32.shape[-1], sg_window_size, sg_poly_order)
    if window is None:
        return np.array(waves_f32, copy=True)
    out = savgol_filter(waves_f32, window_length=window, polyorder=int(sg_poly_order),
                        axis=-1, mode="interp")
    return np.asarray(out, dtype=np.float32)


def filter_wave_pool(records: np.ndarray, wave_pool: np.ndarray, filter_type: str = "SG", *,
                     bw_sos=None, sg_window_size=11, sg_poly_order=2,
                     per_record_cfg=None) -> np.ndarray:
    """records.py:368-438 + filtering.py:377-407: float32 pool aligned to wave_pool, gaps 0.0.

    per_record_cfg: optional callable(i) -> dict overriding the filter kwargs for record i
    (the reference resolves this per hardware channel, filtering.py:339-374).
    """
    out = np.zeros(len(wave_pool), dtype=np.float32)
    for i in range(len(records)):
        length = int(records["event_length"][i])
        if length <= 0:
            continue
        offset = int(records["wave_offset"][i])
        end = offset + length
        if offset < 0 or end > len(wave_pool):
            raise ValueError(
                "wave_pool_filtered found out-of-bounds wave slice "
                f"(offset={offset}, length={length}, wave_pool_size={len(wave_pool)})"
            )
        kw = dict(filter_type=filter_type, bw_sos=bw_sos, sg_window_size=sg_window_size,
                  sg_poly_order=sg_poly_order)
        if per_record_cfg is not None:
            kw.update(per_record_cfg(i))
        ft = kw.pop("filter_type")
        out[offset:end] = apply_filter_core(np.asarray(wave_pool[offset:end], dtype=np.float32), ft, **kw)
    return out


def filter_wave_pool_uniform(wave_pool: np.ndarray, L: int, **kw) -> np.ndarray:
    """Vectorised filter_wave_pool for contiguous equal-length records (pool = R*L samples)."""
    x = np.asarray(wave_pool).reshape(-1, L).astype(np.float32)
    return apply_filter_core(x, kw.pop("filter_type", "SG"), **kw).reshape(-1)


# --- a8: RecordsView padded batch -------------------------------------------------------------
def waves_padded(records: np.ndarray, wave_pool: np.ndarray, dtype=np.float64):
    """records_view.py:208-257 (`rv.waves(ids, mask=True, dtype=...)`): zero-padded matrix + mask."""
    lengths = records["event_length"].astype(np.int64)
    offsets = records["wave_offset"].astype(np.int64)
    if np.any(offsets < 0):
        raise ValueError("records contain negative wave_offset values")
    if np.any(lengths < 0):
        raise ValueError("records contain negative event_length values")
    if len(records) and np.any(offsets + lengths > len(wave_pool)):
        raise ValueError("records reference samples outside wave_pool bounds")
    max_len = int(lengths.max()) if lengths.size else 0
    out = np.zeros((len(records), max_len), dtype=dtype)
    mask = np.zeros((len(records), max_len), dtype=bool)
    for i in range(len(records)):
        n = int(lengths[i])
        if n == 0:
            continue
        out[i, :n] = wave_pool[offsets[i] : offsets[i] + n]
        mask[i, :n] = True
    return out, mask


# --- a6: threshold hits -----------------------------------------------------------------------
def positive_mask_from_polarity(records: np.ndarray) -> np.ndarray:
    """hit_finder.py:322-325: only an explicit "positive" flips the sign."""
    if "polarity" not in (records.dtype.names or ()):
        return np.zeros(len(records), dtype=bool)
    pol = np.asarray(records["polarity"]).astype("U16")
    return pol == "positive"


def hits_from_signal_matrix(signal, thresholds, timestamps, boards, channels, record_ids,
                            left_extension, right_extension, dt_values, valid_mask,
                            record_lengths) -> np.ndarray:
    """hit_finder.py:329-413, literal (per-hit python loop)."""
    if signal.size == 0:
        return np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)
    mask = signal >= thresholds[:, np.newaxis]
    if valid_mask is not None:
        mask &= valid_mask
    if not np.any(mask):
        return np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)
    mask_padded = np.pad(mask, ((0, 0), (1, 1)), mode="constant", constant_values=False)
    diff = np.diff(mask_padded.astype(np.int8), axis=1)
    start_rows, starts = np.where(diff == 1)
    _end_rows, ends = np.where(diff == -1)
    n_samples = signal.shape[1]
    hits = []
    for hit_idx, event_idx in enumerate(start_rows.tolist()):
        start = int(starts[hit_idx])
        end = int(ends[hit_idx])
        seg_start = max(0, start - left_extension)
        seg_end = min(n_samples, end + right_extension)
        if seg_end <= seg_start:
            continue
        segment = signal[event_idx, seg_start:seg_end]
        rel_pos = int(np.argmax(segment))
        pos = seg_start + rel_pos
        height = float(segment[rel_pos])
        integral = float(np.sum(np.maximum(segment, 0.0)))
        dt_ns = int(dt_values[event_idx])
        sampling_interval_ps = float(dt_ns) * 1e3
        rise_time = float(max(pos - start, 0) * dt_ns)
        fall_time = float(max((end - 1) - pos, 0) * dt_ns)
        global_timestamp = int(timestamps[event_idx] + pos * sampling_interval_ps)
        record_length = max(int(record_lengths[event_idx]), 0)
        edge_start = min(max(seg_start, 0), record_length)
        edge_end = min(max(seg_end, 0), record_length)
        edge_end = max(edge_end, edge_start)
        hits.append((int(pos), height, integral, edge_start, edge_end,
                     float(edge_end - edge_start), dt_ns, rise_time, fall_time,
                     global_timestamp, int(boards[event_idx]), int(channels[event_idx]),
                     int(record_ids[event_idx])))
    if hits:
        return np.array(hits, dtype=THRESHOLD_HIT_DTYPE)
    return np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)


def threshold_hits(records: np.ndarray, wave_pool: np.ndarray, *, threshold=10.0,
                   thresholds=None, left_extension=2, right_extension=2) -> np.ndarray:
    """hit_finder.py:122-255 records branch: wave_pool may be uint16 (raw) or float32 (filtered).

    thresholds: optional per-record float64 array (per-channel overrides already resolved,
    hit_finder.py:288-327).
    """
    if len(records) == 0:
        return np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)
    waves, valid = waves_padded(records, wave_pool, dtype=np.float64)
    baselines = records["baseline"].astype(np.float64)
    thr = (np.full(len(records), float(threshold), dtype=np.float64)
           if thresholds is None else np.asarray(thresholds, dtype=np.float64))
    positive = positive_mask_from_polarity(records)
    b2 = baselines[:, np.newaxis]
    signal = np.where(positive[:, np.newaxis], waves - b2, b2 - waves)
    return hits_from_signal_matrix(
        signal, thr, records["timestamp"].astype(np.int64), records["board"].astype(np.int16),
        records["channel"].astype(np.int16), records["record_id"].astype(np.int64),
        max(0, int(left_extension)), max(0, int(right_extension)),
        records["dt"].astype(np.int32), valid, records["event_length"].astype(np.int64))


def threshold_hits_chunked(records, wave_pool, chunk=4096, **kw) -> np.ndarray:
    """threshold_hits over record chunks (bounded memory).

    Only valid when every chunk has the same maximum record length as the whole input
    (the padded-width semantics of hit_finder.py:364,370 depend on it) -- true for the
    equal-length synthetic runs this is used on.
    """
    thresholds = kw.pop("thresholds", None)
    parts = []
    for lo in range(0, len(records), chunk):
        sub_thr = None if thresholds is None else thresholds[lo : lo + chunk]
        parts.append(threshold_hits(records[lo : lo + chunk], wave_pool, thresholds=sub_thr, **kw))
    return np.concatenate(parts) if parts else np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)


# --- a9: basic features -----------------------------------------------------------------------
def _normalized_signal_f32(rec, wave, baseline):
    """records_view.py:87-100: f32(w) - f32(b), negated iff polarity == "positive"."""
    signal = wave.astype(np.float32, copy=False) - np.asarray(baseline, dtype=np.float32)
    if str(rec["polarity"]) == "positive":
        signal = -signal
    return signal


def basic_features(records: np.ndarray, wave_pool: np.ndarray, *, height_range=(40, 90),
                   area_range=(0, None), fixed_baseline=None) -> np.ndarray:
    """basic_features.py:108-195 records branch, literal per-record loop.

    fixed_baseline: optional per-record array, NaN = "use records.baseline"
    (the per-channel override of basic_features.py:133-146 already resolved).
    """
    start_p, end_p = height_range
    start_c, end_c = area_range
    out = np.zeros(len(records), dtype=BASIC_FEATURES_DTYPE)
    has_pol = "polarity" in (records.dtype.names or ())
    for idx, rec in enumerate(records):
        baseline = float(rec["baseline"])
        if fixed_baseline is not None and not np.isnan(fixed_baseline[idx]):
            baseline = float(fixed_baseline[idx])
        off, n = int(rec["wave_offset"]), int(rec["event_length"])
        wave = wave_pool[off : off + n]
        pol = str(rec["polarity"]) if has_pol else None
        use_norm = pol in ("positive", "negative")
        signal = -_normalized_signal_f32(rec, wave, baseline) if use_norm else None
        wave_p, wave_c = wave[start_p:end_p], wave[start_c:end_c]
        signal_p = signal[start_p:end_p] if signal is not None else None
        signal_c = signal[start_c:end_c] if signal is not None else None
        eff = pol if use_norm else "negative"
        if use_norm and signal_p.size > 0:
            s_min, s_max = float(np.min(signal_p)), float(np.max(signal_p))
            out["height"][idx] = s_max
            out["amp"][idx] = s_max - s_min
        elif wave_p.size > 0:
            w_min, w_max = float(np.min(wave_p)), float(np.max(wave_p))
            out["height"][idx] = (w_max - baseline) if eff == "positive" else (baseline - w_min)
            out["amp"][idx] = w_max - w_min
        if use_norm and signal_c.size > 0:
            out["area"][idx] = float(np.sum(signal_c.astype(np.float64, copy=False)))
        elif wave_c.size > 0:
            wave_c64 = wave_c.astype(np.float64)
            b64 = np.asarray(baseline, dtype=np.float64)
            out["area"][idx] = float(np.sum(wave_c64 - b64)) if eff == "positive" else float(np.sum(b64 - wave_c64))
        if wave.size > 1:
            out["max_abs_diff"][idx] = float(np.max(np.abs(np.diff(wave.astype(np.float64, copy=False)))))
        out["timestamp"][idx] = int(rec["timestamp"])
        out["board"][idx] = int(rec["board"])
        out["channel"][idx] = int(rec["channel"])
        out["event_index"][idx] = idx
    return out


# --- a10: integral-quantile width -------------------------------------------------------------
def width_integral(records: np.ndarray, wave_pool: np.ndarray, *, q_low=0.10, q_high=0.90,
                   dt=None, sampling_rate=0.5) -> np.ndarray:
    """waveform_width_integral.py:83-231 records branch, literal per-record loop."""
    if dt is None:
        dt = 1.0 / float(sampling_rate)
    has_pol = "polarity" in (records.dtype.names or ())
    rows = []
    for idx, rec in enumerate(records):
        off, n = int(rec["wave_offset"]), int(rec["event_length"])
        wave = wave_pool[off : off + n]
        baseline = float(rec["baseline"])
        pol = str(rec["polarity"]) if has_pol else "unknown"
        if pol in ("positive", "negative"):
            signal = -_normalized_signal_f32(rec, wave, rec["baseline"]).astype(np.float64, copy=False)
        else:
            raw_signal = wave.astype(np.float64, copy=False) - baseline
            signal = raw_signal if pol == "positive" else -raw_signal
        x = np.maximum(signal, 0.0)
        q_total = float(np.sum(x))
        if q_total <= 0 or not np.isfinite(q_total):
            lo = hi = w = 0.0
        else:
            cumsum = np.cumsum(x)
            lo_i = int(np.searchsorted(cumsum, q_low * q_total, side="left"))
            hi_i = int(np.searchsorted(cumsum, q_high * q_total, side="left"))
            lo, hi, w = float(lo_i), float(hi_i), float(max(hi_i - lo_i, 0))
        rows.append((float(lo * dt), float(hi * dt), float(w * dt), lo, hi, w, q_total,
                     int(rec["timestamp"]), int(rec["board"]), int(rec["channel"]), idx))
    if rows:
        return np.array(rows, dtype=WAVEFORM_WIDTH_INTEGRAL_DTYPE)
    return np.zeros(0, dtype=WAVEFORM_WIDTH_INTEGRAL_DTYPE)


# --- a15: event grouping ----------------------------------------------------------------------
def group_hit_windows_literal(hits: np.ndarray, time_window_ns: float):
    """event_grouping.py:286-471, literal gap-chain loop; returns a list of events
    (t_min, t_max, member hit indices in the reference's per-event order)."""
    names = set(hits.dtype.names or ())
    s_name, e_name = ("sample_start", "sample_end") if {"sample_start", "sample_end"} <= names else ("edge_start", "edge_end")
    ts = np.asarray(hits["timestamp"], dtype=np.int64)
    pos = np.asarray(hits["position"], dtype=np.float64)
    dt = np.asarray(hits["dt"], dtype=np.int32)
    s_rel = np.asarray(hits[s_name], dtype=np.int32)
    e_rel = np.asarray(hits[e_name], dtype=np.int32)
    rid = np.asarray(hits["record_id"], dtype=np.int64)
    boards = np.asarray(hits["board"], dtype=np.int16)
    channels = np.asarray(hits["channel"], dtype=np.int16)
    dt_ps = dt.astype(np.float64) * 1e3
    a_s = ts.astype(np.float64) + (s_rel - pos) * dt_ps
    a_e = ts.astype(np.float64) + (e_rel - pos) * dt_ps
    if len(hits) == 0:
        return []
    order = np.lexsort((rid, ts, dt, a_s))
    gap_ps = time_window_ns * 1e3

    def build(idxs):
        sub = np.asarray(idxs, dtype=np.int64)
        k = np.lexsort((rid[sub], ts[sub], a_s[sub], dt[sub], channels[sub], boards[sub]))
        sub = sub[k]
        return int(np.min(a_s[sub])), int(np.max(a_e[sub])), sub

    events = []
    cur = [int(order[0])]
    cluster_end = float(a_e[order[0]])
    for idx in order[1:]:
        idx = int(idx)
        if a_s[idx] <= cluster_end + gap_ps:
            cur.append(idx)
            cluster_end = max(cluster_end, float(a_e[idx]))
        else:
            events.append(build(cur))
            cur = [idx]
            cluster_end = float(a_e[idx])
    events.append(build(cur))
    return events


# --- a11: find_peaks-based hit detector -------------------------------------------------------
HIT_DTYPE = np.dtype(
    [
        ("position", "i8"), ("height", "f4"), ("integral", "f4"), ("edge_start", "f4"), ("edge_end", "f4"),
        ("dt", "i4"), ("timestamp", "i8"), ("board", "i2"), ("channel", "i2"), ("record_id", "i8"),
    ]
)


def _select_by_peak_distance(peaks: np.ndarray, priority: np.ndarray, distance: float) -> np.ndarray:
    """scipy/signal/_peak_finding_utils.pyx `_select_by_peak_distance` (scipy 1.15), restated.

    scipy orders the peaks with np.argsort(priority) (numpy's default introsort, whose order of EQUAL
    priorities is an implementation detail: insertion sort -- stable -- up to 16 elements, vectorised
    sorts above that on some CPUs).  The restatement pins the stable order, so among equal priorities
    the LATER peak is visited first; without ties the result is scipy's on every platform.
    """
    n = peaks.shape[0]
    distance_ = math.ceil(distance)
    keep = np.ones(n, dtype=bool)
    order = np.argsort(priority, kind="stable")
    for i in range(n - 1, -1, -1):
        j = order[i]
        if not keep[j]:
            continue
        k = j - 1
        while 0 <= k and peaks[j] - peaks[k] < distance_:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < n and peaks[k] - peaks[j] < distance_:
            keep[k] = False
            k += 1
    return keep


def find_peaks_staged(x: np.ndarray, height: float, threshold, distance: int, prominence: float, width: float):
    """scipy.signal.find_peaks(x, height, threshold, distance, prominence, width) with scalar lower
    bounds (scipy/signal/_peak_finding.py `find_peaks`, the order of its condition blocks), built from
    scipy's own public pieces except the distance step (see _select_by_peak_distance)."""
    from scipy.signal import find_peaks, peak_prominences, peak_widths

    if distance is not None and distance < 1:
        raise ValueError("`distance` must be greater or equal to 1")
    peaks, _ = find_peaks(x)  # local maxima incl. plateau midpoints (_local_maxima_1d)
    peaks = peaks[x[peaks] >= height]
    if threshold is not None:
        lt, rt = x[peaks] - x[peaks - 1], x[peaks] - x[peaks + 1]
        peaks = peaks[np.minimum(lt, rt) >= threshold]
    if distance is not None:
        peaks = peaks[_select_by_peak_distance(peaks, x[peaks], distance)]
    prom, lb, rb = peak_prominences(x, peaks, wlen=None)
    ok = prom >= prominence
    peaks, prom, lb, rb = peaks[ok], prom[ok], lb[ok], rb[ok]
    widths, _wh, l_ips, r_ips = peak_widths(x, peaks, 0.5, (prom, lb, rb), None)
    ok = widths >= width
    return peaks[ok], l_ips[ok], r_ips[ok]


def find_peak_hits(records: np.ndarray, wave_pool: np.ndarray, *, use_derivative=True, height=30.0,
                   distance=2, prominence=0.7, width=4, threshold=None, height_method="minmax",
                   height_window_extension=4) -> np.ndarray:
    """peak_finding.py:395-614, records branch (`_process_records_range` + `_find_peaks_in_waveform`
    + `_calculate_peak_height`), literal per-record loop around find_peaks_staged."""
    if height_method not in ("minmax", "diff"):
        raise ValueError(f"不支持的峰高计算方法: {height_method}")
    rows = []
    for rec in records:
        off, n = int(rec["wave_offset"]), int(rec["event_length"])
        wave = wave_pool[off : off + n]
        signal = -_normalized_signal_f32(rec, wave, rec["baseline"]).astype(np.float64, copy=False)
        if signal.size == 0:
            continue
        det = np.diff(signal) if use_derivative else signal - 0.0
        peaks, l_ips, r_ips = find_peaks_staged(det, float(height), threshold, int(distance), float(prominence),
                                                int(width))
        dt_ns = int(rec["dt"])
        for pos, l_ip, r_ip in zip(peaks, l_ips, r_ips):
            start_idx = max(0, int(np.round(l_ip)))
            end_idx = min(len(signal) - 1, int(np.round(r_ip)))
            if height_method == "minmax":
                ext = max(0, int(height_window_extension))
                w0, w1 = max(0, start_idx - ext), min(len(signal), end_idx + ext)
                ph = np.max(signal[w0:w1]) - np.min(signal[w0:w1])
            else:
                ph = np.sum(np.diff(-signal)[start_idx:end_idx]) if end_idx > start_idx else 0.0
            ts = int(int(rec["timestamp"]) + pos * (dt_ns * 1e3))
            rows.append((int(pos), float(ph), 0.0, float(l_ip), float(r_ip), dt_ns, ts, int(rec["board"]),
                         int(rec["channel"]), int(rec["record_id"])))
    return np.array(rows, dtype=HIT_DTYPE) if rows else np.zeros(0, dtype=HIT_DTYPE)


# ------------------------------------------------------------------------------------------------
# Dense (st_waveforms / filtered_waveforms) branches, waveform_width, s1_s2
# ------------------------------------------------------------------------------------------------
WAVEFORM_WIDTH_DTYPE = np.dtype(
    [("rise_time", "f4"), ("fall_time", "f4"), ("total_width", "f4"), ("rise_time_samples", "f4"),
     ("fall_time_samples", "f4"), ("total_width_samples", "f4"), ("peak_position", "i8"), ("peak_height", "f4"),
     ("timestamp", "i8"), ("board", "i2"), ("channel", "i2"), ("record_id", "i8")]
)
S1_S2_CLASSIFIER_DTYPE = np.dtype(
    [("label", "i1"), ("width_ns", "f4"), ("width_samples", "f4"), ("height", "f4"), ("area", "f4"),
     ("timestamp", "i8"), ("board", "i2"), ("channel", "i2"), ("record_id", "i8"), ("peak_position", "i8")]
)


def filtered_waveforms_dense(st: np.ndarray, cfg_of_channel) -> np.ndarray:
    """filtering.py:444-536: each hardware channel's rows are filtered as ONE 2-D float32 batch (axis=-1).

    cfg_of_channel(board, channel) -> kwargs of apply_filter_core.  Returns the float32 wave matrix."""
    waves = st["wave"]
    out = np.empty(waves.shape, dtype=np.float32)
    boards = st["board"] if "board" in st.dtype.names else np.zeros(len(st), dtype=np.int16)
    keys = sorted(set(zip(boards.tolist(), st["channel"].tolist())))
    for b, c in keys:
        rows = np.flatnonzero((boards == b) & (st["channel"] == c))
        out[rows] = apply_filter_core(np.asarray(waves[rows], dtype=np.float32), **cfg_of_channel(b, c))
    return out


def basic_features_dense(data: np.ndarray, *, height_range=(40, 90), area_range=(0, None),
                         fixed_baseline=None) -> np.ndarray:
    """basic_features.py:197-278 (st_waveforms / filtered_waveforms): whole rows, wave-based formulas for every
    polarity, sign flipped only for the literal "positive"."""
    p0, p1 = height_range
    c0, c1 = area_range
    out = np.zeros(len(data), dtype=BASIC_FEATURES_DTYPE)
    names = data.dtype.names
    for i in range(len(data)):
        wave = data["wave"][i]
        b = float(data["baseline"][i])
        if fixed_baseline is not None and not np.isnan(fixed_baseline[i]):
            b = float(fixed_baseline[i])
        positive = "polarity" in names and str(data["polarity"][i]) == "positive"
        wp = wave[p0:p1]
        if wp.size:
            lo, hi = float(np.min(wp)), float(np.max(wp))
            out["height"][i] = (hi - b) if positive else (b - lo)
            out["amp"][i] = hi - lo
        wc = wave[c0:c1].astype(np.float64, copy=False)
        if wc.size:
            out["area"][i] = float(np.sum(wc - b)) if positive else float(np.sum(b - wc))
        if wave.size > 1:
            out["max_abs_diff"][i] = float(np.max(np.abs(np.diff(wave.astype(np.float64, copy=False)))))
    out["timestamp"] = data["timestamp"]
    out["board"] = data["board"] if "board" in names else 0
    out["channel"] = data["channel"] if "channel" in names else 0
    out["event_index"] = np.arange(len(data))
    return out


def threshold_hits_dense(data: np.ndarray, record_lengths: np.ndarray, *, threshold=10.0, thresholds=None,
                         left_extension=2, right_extension=2) -> np.ndarray:
    """hit_finder.py:179-255 dense branch: the WHOLE row is searched (no valid mask), `record_lengths` (the
    records/wave_pool lengths of the same record_ids, hit_finder.py:257-286) only clamp the edges."""
    if len(data) == 0:
        return np.zeros(0, dtype=THRESHOLD_HIT_DTYPE)
    names = data.dtype.names or ()
    n = len(data)
    waves = np.asarray(data["wave"]).astype(np.float64, copy=False)
    baselines = data["baseline"].astype(np.float64) if "baseline" in names else waves.mean(axis=1, dtype=np.float64)
    thr = (np.full(n, float(threshold), dtype=np.float64) if thresholds is None
           else np.asarray(thresholds, dtype=np.float64))
    positive = positive_mask_from_polarity(data)
    b2 = baselines[:, np.newaxis]
    signal = np.where(positive[:, np.newaxis], waves - b2, b2 - waves)
    col = lambda f, t, d: data[f].astype(t) if f in names else d  # noqa: E731
    return hits_from_signal_matrix(
        signal, thr, col("timestamp", np.int64, np.zeros(n, np.int64)), col("board", np.int16, np.zeros(n, np.int16)),
        col("channel", np.int16, np.zeros(n, np.int16)), col("record_id", np.int64, np.arange(n, dtype=np.int64)),
        max(0, int(left_extension)), max(0, int(right_extension)), data["dt"].astype(np.int32), None,
        np.asarray(record_lengths, dtype=np.int64))


def width_integral_dense(data: np.ndarray, *, q_low=0.10, q_high=0.90, dt=None, sampling_rate=0.5) -> np.ndarray:
    """waveform_width_integral.py:139-231 dense branch: float64 of the row minus the float64 baseline, sign from
    the literal "positive" only."""
    if dt is None:
        dt = 1.0 / float(sampling_rate)
    names = data.dtype.names or ()
    rows = []
    for idx in range(len(data)):
        wave = data[idx]["wave"]
        baseline = float(data[idx]["baseline"])
        pol = str(data[idx]["polarity"]) if "polarity" in names else "unknown"
        raw_signal = wave.astype(np.float64, copy=False) - baseline
        signal = raw_signal if pol == "positive" else -raw_signal
        x = np.maximum(signal, 0.0)
        q_total = float(np.sum(x))
        if q_total <= 0 or not np.isfinite(q_total):
            lo = hi = w = 0.0
        else:
            cumsum = np.cumsum(x)
            lo_i = int(np.searchsorted(cumsum, q_low * q_total, side="left"))
            hi_i = int(np.searchsorted(cumsum, q_high * q_total, side="left"))
            lo, hi, w = float(lo_i), float(hi_i), float(max(hi_i - lo_i, 0))
        rows.append((float(lo * dt), float(hi * dt), float(w * dt), lo, hi, w, q_total, int(data[idx]["timestamp"]),
                     int(data[idx]["board"]) if "board" in names else 0,
                     int(data[idx]["channel"]) if "channel" in names else 0, idx))
    if rows:
        return np.array(rows, dtype=WAVEFORM_WIDTH_INTEGRAL_DTYPE)
    return np.zeros(0, dtype=WAVEFORM_WIDTH_INTEGRAL_DTYPE)


def find_peak_hits_dense(data: np.ndarray, *, use_derivative=True, height=30.0, distance=2, prominence=0.7, width=4,
                         threshold=None, height_method="minmax", height_window_extension=4) -> np.ndarray:
    """peak_finding.py:316-378 + 446-614 dense branch: the row itself (int16 or float32, truncated to
    event_length) is the waveform, pulses are negative-going (`-np.diff(w)` in the row's dtype, or
    `baseline - w`), the height is taken on the row."""
    if height_method not in ("minmax", "diff"):
        raise ValueError(f"不支持的峰高计算方法: {height_method}")
    names = data.dtype.names or ()
    rows = []
    for idx in range(len(data)):
        row = data[idx]
        w = row["wave"]
        ev = int(row["event_length"]) if "event_length" in names else len(w)
        if 0 < ev < len(w):
            w = w[:ev]
        baseline = row["baseline"] if "baseline" in names else None
        if use_derivative:
            det = -np.diff(w)
        elif baseline is not None:
            det = baseline - w
        else:
            det = np.mean(w) - w
        det = np.asarray(det, dtype=np.float64)
        peaks, l_ips, r_ips = find_peaks_staged(det, float(height), threshold, int(distance), float(prominence),
                                                int(width))
        dt_ns = int(row["dt"])
        for pos, l_ip, r_ip in zip(peaks, l_ips, r_ips):
            start_idx = max(0, int(np.round(l_ip)))
            end_idx = min(len(w) - 1, int(np.round(r_ip)))
            if height_method == "minmax":
                ext = max(0, int(height_window_extension))
                w0, w1 = max(0, start_idx - ext), min(len(w), end_idx + ext)
                ph = np.max(w[w0:w1]) - np.min(w[w0:w1])
            else:
                ph = np.sum(np.diff(-w)[start_idx:end_idx]) if end_idx > start_idx else 0.0
            ts = int(row["timestamp"] + pos * (dt_ns * 1e3))
            rows.append((int(pos), float(ph), 0.0, float(l_ip), float(r_ip), dt_ns, ts,
                         int(row["board"]) if "board" in names else 0, int(row["channel"]) if "channel" in names else 0,
                         int(row["record_id"]) if "record_id" in names else idx))
    return np.array(rows, dtype=HIT_DTYPE) if rows else np.zeros(0, dtype=HIT_DTYPE)


def signal_peaks_rows(st_rows: np.ndarray, filtered_rows: np.ndarray, *, use_derivative=True, height=30.0,
                      distance=2, prominence=0.7, width=4, threshold=None, height_method="diff",
                      minmax_window_expand=2, explicit_dt=None, event_offset=0) -> np.ndarray:
    """streaming/cpu/signal_peaks.py:226-406 (`compute_chunk` + `_find_peaks_in_waveform` +
    `_calculate_peak_heights`): peaks of one chunk of rows; the filtered row is converted to float64 first."""
    names = st_rows.dtype.names or ()
    rows = []
    for local_idx, (frow, srow) in enumerate(zip(filtered_rows, st_rows)):
        w = np.asarray(frow["wave"] if getattr(frow, "dtype", None) is not None and frow.dtype.names else frow,
                       dtype=np.float64)
        baseline = srow["baseline"] if "baseline" in names else None
        dt_ns = int(srow["dt"]) if "dt" in names else int(explicit_dt)
        if use_derivative:
            det = -np.diff(w)
        elif baseline is not None:
            det = baseline - w
        else:
            det = np.mean(w) - w
        peaks, l_ips, r_ips = find_peaks_staged(det, height, threshold, distance, prominence, width)
        if len(peaks) == 0:
            continue
        if height_method == "diff":
            d = -np.diff(w)
            cs = np.concatenate(([0.0], np.cumsum(d, dtype=np.float64)))
            s_i = np.clip(np.rint(l_ips).astype(np.int64), 0, len(d))
            e_i = np.clip(np.rint(r_ips).astype(np.int64), 0, len(d))
            heights = np.where(e_i > s_i, cs[e_i] - cs[s_i], 0.0).astype(np.float32)
        elif height_method == "minmax":
            heights = np.zeros(len(peaks), dtype=np.float32)
            for i, (l_ip, r_ip) in enumerate(zip(l_ips, r_ips)):
                s_i, e_i = max(0, int(np.round(l_ip))), min(len(w) - 1, int(np.round(r_ip)))
                w0, w1 = max(0, s_i - minmax_window_expand), min(len(w), e_i + minmax_window_expand)
                heights[i] = np.max(w[w0:w1]) - np.min(w[w0:w1])
        else:
            raise ValueError(f"不支持的峰高计算方法: {height_method}")
        if dt_ns <= 0:
            raise ValueError("[signal_peaks_stream] dt must be > 0")
        for pos, l_ip, r_ip, ph in zip(peaks, l_ips, r_ips, heights):
            ts = int(int(srow["timestamp"]) + pos * (float(dt_ns) * 1e3))
            rows.append((int(pos), float(ph), 0.0, float(l_ip), float(r_ip), dt_ns, ts,
                         int(srow["board"]) if "board" in names else 0, int(srow["channel"]),
                         int(srow["record_id"]) if "record_id" in names else event_offset + local_idx))
    return np.array(rows, dtype=HIT_DTYPE) if rows else np.zeros(0, dtype=HIT_DTYPE)


def _first_crossing(seg: np.ndarray, level, rising: bool, interpolate: bool):
    """waveform_width.py:332-374: first sample at/over (rising) or at/under (falling) `level`, refined by the
    straight line through its left neighbour.  Keeps numpy's scalar types exactly as the reference's
    expressions produce them (python float for the un-refined index)."""
    if seg.size == 0:
        return None
    hit = np.flatnonzero(seg >= level) if rising else np.flatnonzero(seg <= level)
    if hit.size == 0:
        return None
    k = hit[0]
    if not interpolate or k == 0:
        return float(k)
    left, right = seg[k - 1], seg[k]
    if abs(right - left) < 1e-10:
        return float(k)
    return float(k - 1) + (level - left) / (right - left)


def waveform_width(hits: np.ndarray, waveform_data: np.ndarray, *, rise_low=0.1, rise_high=0.9, fall_high=0.9,
                   fall_low=0.1, sampling_rate=0.5, interpolation=True) -> np.ndarray:
    """waveform_width.py:139-330, literal per-hit loop (row found by the FIRST matching record_id)."""
    rows = []
    has_rid = "record_id" in (waveform_data.dtype.names or ())
    for h in hits:
        rid = int(h["record_id"]) if "record_id" in h.dtype.names else int(h["event_index"])
        if has_rid:
            where = np.flatnonzero(waveform_data["record_id"] == rid)
            if where.size == 0:
                continue
            wave = waveform_data[int(where[0])]["wave"]
        else:
            if not 0 <= rid < len(waveform_data):
                continue
            wave = waveform_data[rid]["wave"]
        pos = h["position"]
        corrected = wave - np.mean(wave[:50])
        if pos >= len(corrected):
            continue
        top = corrected[pos]
        if top <= 0:
            continue
        before, after = corrected[:pos], corrected[pos:]
        r_lo = _first_crossing(before, top * rise_low, True, interpolation)
        r_hi = _first_crossing(before, top * rise_high, True, interpolation)
        f_hi = _first_crossing(after, top * fall_high, False, interpolation)
        f_lo = _first_crossing(after, top * fall_low, False, interpolation)
        rise_s = rise_t = fall_s = fall_t = tot_s = tot_t = 0.0
        if r_lo is not None and r_hi is not None:
            rise_s = r_hi - r_lo
            rise_t = rise_s / sampling_rate
        if f_hi is not None and f_lo is not None:
            f_hi = f_hi + pos  # np.int64 position: promotes to float64
            f_lo = f_lo + pos
            fall_s = f_lo - f_hi
            fall_t = fall_s / sampling_rate
        if r_lo is not None and f_lo is not None:
            tot_s = f_lo - r_lo
            tot_t = tot_s / sampling_rate
        rows.append((float(rise_t), float(fall_t), float(tot_t), float(rise_s), float(fall_s), float(tot_s), int(pos),
                     float(top), int(h["timestamp"]), int(h["board"]) if "board" in h.dtype.names else 0,
                     int(h["channel"]), rid))
    return np.array(rows, dtype=WAVEFORM_WIDTH_DTYPE) if rows else np.zeros(0, dtype=WAVEFORM_WIDTH_DTYPE)


def s1_s2_classify(widths: np.ndarray, features: np.ndarray, *, width_unit="ns", s1_width_range=None,
                   s2_width_range=None, s1_area_range=None, s2_area_range=None, s1_height_range=None,
                   s2_height_range=None, conflict_policy="unknown", strict=False) -> np.ndarray:
    """s1_s2_classifier.py:133-228, literal per-row loop."""

    def norm(r):
        if r is None:
            return None
        if not isinstance(r, tuple) or len(r) != 2:
            raise ValueError("range must be a tuple of (min, max)")
        if r[0] is None and r[1] is None:
            return None
        return (None if r[0] is None else float(r[0]), None if r[1] is None else float(r[1]))

    def inside(v, r):
        if r is None:
            return True
        if v is None or np.isnan(v):
            return False
        return not ((r[0] is not None and v < r[0]) or (r[1] is not None and v > r[1]))

    s1 = [norm(s1_width_range), norm(s1_area_range), norm(s1_height_range)]
    s2 = [norm(s2_width_range), norm(s2_area_range), norm(s2_height_range)]
    s1_on, s2_on = any(r is not None for r in s1), any(r is not None for r in s2)
    if strict and not s1_on and not s2_on:
        raise ValueError("No S1/S2 criteria configured; set ranges or disable strict.")
    rows = []
    for w in widths:
        rid = int(w["record_id"]) if "record_id" in w.dtype.names else int(w["event_index"])
        h = a = np.nan
        if "record_id" in (features.dtype.names or ()):
            m = np.flatnonzero(features["record_id"] == rid)
            if m.size:
                h, a = float(features["height"][m[0]]), float(features["area"][m[0]])
        elif 0 <= rid < len(features):
            h, a = float(features["height"][rid]), float(features["area"][rid])
        wn, ws = float(w["total_width"]), float(w["total_width_samples"])
        wv = ws if width_unit == "samples" else wn
        ok1 = s1_on and inside(wv, s1[0]) and inside(a, s1[1]) and inside(h, s1[2])
        ok2 = s2_on and inside(wv, s2[0]) and inside(a, s2[1]) and inside(h, s2[2])
        if ok1 and ok2:
            label = {"prefer_s1": 1, "prefer_s2": 2}.get(conflict_policy, 0)
        else:
            label = 1 if ok1 else (2 if ok2 else 0)
        rows.append((label, wn, ws, h, a, int(w["timestamp"]), int(w["board"]) if "board" in w.dtype.names else 0,
                     int(w["channel"]), rid, int(w["peak_position"])))
    return np.array(rows, dtype=S1_S2_CLASSIFIER_DTYPE) if rows else np.zeros(0, dtype=S1_S2_CLASSIFIER_DTYPE)


# ------------------------------------------------------------------------------------------------
# Hit merging (cpu/hit_merge.py), literal loops over python lists as the reference walks them
# ------------------------------------------------------------------------------------------------
HIT_MERGED_DTYPE = np.dtype(
    [("position", "i8"), ("height", "f4"), ("integral", "f4"), ("sample_start", "i4"), ("sample_end", "i4"),
     ("width", "f4"), ("dt", "i4"), ("rise_time", "f4"), ("fall_time", "f4"), ("timestamp", "i8"), ("board", "i2"),
     ("channel", "i2"), ("record_id", "i8"), ("component_offset", "i8"), ("component_count", "i4")]
)
HIT_MERGE_CLUSTERS_DTYPE = np.dtype([("cluster_index", "i8"), ("hit_index", "i8")])


def hit_merge_clusters(hits: np.ndarray, merge_gap_ns: float = 0.0, max_total_width_ns: float = 10000.0) -> list:
    """hit_merge.py:115-181: list of clusters, each a list of hit indices in chain order."""
    if len(hits) == 0:
        return []
    boards = hits["board"].astype(np.int64) if "board" in hits.dtype.names else np.zeros(len(hits), dtype=np.int64)
    channels = hits["channel"].astype(np.int64)
    dt_ps = hits["dt"].astype(np.float64) * 1e3
    abs0 = hits["timestamp"].astype(np.float64) + (hits["edge_start"].astype(np.float64) - hits["position"].astype(np.float64)) * dt_ps
    abs1 = hits["timestamp"].astype(np.float64) + (hits["edge_end"].astype(np.float64) - hits["position"].astype(np.float64)) * dt_ps
    gap_ps, cap_ps = merge_gap_ns * 1e3, max_total_width_ns * 1e3
    clusters = []
    for b, c in sorted(set(zip(boards.tolist(), channels.tolist()))):
        idx = np.flatnonzero((boards == b) & (channels == c))
        idx = idx[np.argsort(abs0[idx], kind="mergesort")]
        current = [int(idx[0])]
        c_start, c_end = abs0[idx[0]], abs1[idx[0]]
        for i in idx[1:]:
            i = int(i)
            nxt = max(c_end, abs1[i])
            if (merge_gap_ns > 0 and dt_ps[i] == dt_ps[current[-1]] and abs0[i] - c_end <= gap_ps
                    and nxt - c_start <= cap_ps):
                current.append(i)
                c_end = nxt
            else:
                clusters.append(current)
                current = [i]
                c_start, c_end = abs0[i], abs1[i]
        clusters.append(current)
    return clusters


def hit_merge_cluster_rows(clusters: list) -> np.ndarray:
    rows = [(k, i) for k, members in enumerate(clusters) for i in members]
    return np.array(rows, dtype=HIT_MERGE_CLUSTERS_DTYPE) if rows else np.zeros(0, dtype=HIT_MERGE_CLUSTERS_DTYPE)


def hit_merged_rows(hits: np.ndarray, clusters: list) -> np.ndarray:
    """hit_merge.py:256-322 per cluster."""
    rows = []
    offset = 0
    for members in clusters:
        sub = hits[members]
        if len(members) == 1:
            h = sub[0]
            rows.append((int(h["position"]), float(h["height"]), float(h["integral"]), int(h["edge_start"]),
                         int(h["edge_end"]), float(h["width"]), int(h["dt"]), float(h["rise_time"]), float(h["fall_time"]),
                         int(h["timestamp"]), int(h["board"]), int(h["channel"]), int(h["record_id"]), offset, 1))
            offset += 1
            continue
        heights = sub["height"].astype(np.float64)
        top = float(np.max(heights))
        tied = [k for k in range(len(members)) if float(heights[k]) == top]
        a = sub[min(tied, key=lambda k: int(sub["timestamp"][k]))]
        if len(set(sub["record_id"].tolist())) == 1:
            s0, s1 = int(sub["edge_start"].min()), int(sub["edge_end"].max())
        else:
            s0 = s1 = -1
        width = -1.0 if (s0 < 0 or s1 < 0) else float(max(s1 - s0, 0.0))
        total = float(np.sum([float(v) for v in sub["integral"]]))
        rows.append((int(a["position"]), top, total, s0, s1, width, int(a["dt"]), float(a["rise_time"]),
                     float(a["fall_time"]), int(a["timestamp"]), int(a["board"]), int(a["channel"]), int(a["record_id"]),
                     offset, len(members)))
        offset += len(members)
    return np.array(rows, dtype=HIT_MERGED_DTYPE) if rows else np.zeros(0, dtype=HIT_MERGED_DTYPE)


# ------------------------------------------------------------------------------------------------
# Records builder (processing/records_builder.py), literal restatement
# ------------------------------------------------------------------------------------------------
def records_sort_order(records: np.ndarray) -> np.ndarray:
    """records_builder.py:115-120."""
    seq = np.arange(len(records), dtype=np.int64)
    return np.lexsort((seq, records["channel"], records["board"], records["pid"], records["timestamp"]))


def build_records_from_st_waveforms(st: np.ndarray, default_dt_ns: int = 1):
    """records_builder.py:645-794: rows grouped per hardware channel, concatenated, sorted, waves copied one by one."""
    names = st.dtype.names
    keys = sorted(set(zip(st["board"].tolist(), st["channel"].tolist())))
    grouped = np.concatenate([np.flatnonzero((st["board"] == b) & (st["channel"] == c)) for b, c in keys])
    src = st[grouped]
    rec = np.zeros(len(src), dtype=RECORDS_DTYPE)
    for f in ("timestamp", "channel", "board", "baseline"):
        rec[f] = src[f]
    rec["baseline_upstream"] = src["baseline_upstream"] if "baseline_upstream" in names else np.nan
    rec["polarity"] = src["polarity"] if "polarity" in names else "unknown"
    rec["event_length"] = src["event_length"] if "event_length" in names else src["wave"].shape[1]
    rec["dt"] = src["dt"] if "dt" in names else default_dt_ns
    rec["time"] = src["time"] if "time" in names else rec["timestamp"] // 1000
    order = records_sort_order(rec)
    rec, src = rec[order], src[order]
    if "record_id" in names and np.all(src["record_id"] >= 0):
        rec["record_id"] = src["record_id"]
    else:
        rec["record_id"] = np.arange(len(rec))
    width = src["wave"].shape[1]
    chunks, cursor = [], 0
    for i in range(len(rec)):
        n = min(max(int(rec["event_length"][i]), 0), width)
        rec["event_length"][i] = n
        rec["wave_offset"][i] = cursor
        chunks.append(src["wave"][i][:n].astype(np.uint16))
        cursor += n
    pool = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint16)
    return rec, pool


def merge_records_parts(parts):
    """records_builder.py:869-945: heap k-way merge of sorted (records, wave_pool) parts."""
    import heapq

    total = sum(len(r) for r, _ in parts)
    out = np.zeros(total, dtype=RECORDS_DTYPE)
    heap = []
    for p, (r, _) in enumerate(parts):
        if len(r):
            heapq.heappush(heap, (int(r["timestamp"][0]), int(r["pid"][0]), int(r["board"][0]), int(r["channel"][0]), p, 0))
    chunks, cursor, k = [], 0, 0
    while heap:
        *_, p, row = heapq.heappop(heap)
        r, pool = parts[p]
        n = max(int(r["event_length"][row]), 0)
        off = int(r["wave_offset"][row])
        chunks.append(np.asarray(pool[off : off + n], dtype=np.uint16))
        out[k] = r[row]
        out["wave_offset"][k] = cursor
        cursor += n
        k += 1
        if row + 1 < len(r):
            heapq.heappush(heap, (int(r["timestamp"][row + 1]), int(r["pid"][row + 1]), int(r["board"][row + 1]),
                                  int(r["channel"][row + 1]), p, row + 1))
    if len(np.unique(out["record_id"])) != len(out):
        out["record_id"] = np.arange(total)
    return out, (np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint16))


# ------------------------------------------------------------------------------------------------
# Legacy helpers of processing/event_grouping.py
# ------------------------------------------------------------------------------------------------
PEAK_DTYPE = np.dtype(
    [("time", "i8"), ("area", "f4"), ("height", "f4"), ("width", "f4"), ("channel", "i2"), ("event_index", "i8")]
)


def find_hits_legacy(waves: np.ndarray, baselines: np.ndarray, threshold: float) -> np.ndarray:
    """event_grouping.py:46-95, row by row: a hit per run of (baseline - wave) > threshold, `time` = its first sample."""
    rows = []
    for ev in range(len(waves)):
        above = (baselines[ev] - waves[ev]) > threshold
        prev = False
        for i, m in enumerate(above):
            if m and not prev:
                rows.append((i, 0.0, 0.0, 0.0, 0, ev))
            prev = bool(m)
    return np.array(rows, dtype=PEAK_DTYPE) if rows else np.zeros(0, dtype=PEAK_DTYPE)


def group_multi_channel_hits_literal(ts, ch, area, height, time_window_ns: float):
    """event_grouping.py:98-283 (single process): stable sort by timestamp, windows from each cluster's first hit,
    members stably ordered by channel; returns [(t_min, t_max, members as indices into the input)]."""
    order = np.argsort(ts, kind="stable")
    ts_s = ts[order]
    win = time_window_ns * 1e3
    events, cur, n = [], 0, len(ts_s)
    while cur < n:
        nxt = int(np.searchsorted(ts_s, ts_s[cur] + win, side="right"))
        members = order[cur:nxt]
        members = members[np.argsort(ch[members], kind="stable")]
        events.append((int(ts[members[0]]), int(ts[members[-1]]), members))
        cur = nxt
    return events


def v1725_waves(blob: bytes):
    """utils/formats/v1725.py:66-114: (channel, timestamp, trunc, baseline, samples) of every complete wave."""
    pos, n, out = 0, len(blob), []
    while n - pos >= 16:
        head = blob[pos : pos + 16]
        pos += 16
        mask = head[4] | (head[11] << 8)
        for ch in range(16):
            if not (mask >> ch) & 1:
                continue
            if n - pos < 12:
                return out
            h = blob[pos : pos + 12]
            pos += 12
            words = int.from_bytes(h[:3], "little") & 0x3FFFFF
            nbytes = (words - 3) << 2
            if n - pos < nbytes:
                return out
            out.append((ch, int.from_bytes(h[4:10], "little"), (h[3] >> 6) & 1, int.from_bytes(h[10:12], "little"),
                        np.frombuffer(blob[pos : pos + nbytes], dtype=np.int16)))
            pos += nbytes
    return out


def build_records_from_v1725_blobs(blobs, boards, dt_ns: int):
    """records_builder.py:164-209 + 797-830: one sorted part per file (python loop over waves), heap merge."""
    parts = []
    for blob, board in zip(blobs, boards):
        waves = v1725_waves(bytes(blob))
        if not waves:
            continue
        rec = np.zeros(len(waves), dtype=RECORDS_DTYPE)
        for i, (ch, ts, trunc, bl, w) in enumerate(waves):
            rec[i]["timestamp"] = ts * dt_ns * 1000
            rec[i]["board"], rec[i]["channel"], rec[i]["baseline"] = board, ch, bl
            rec[i]["baseline_upstream"], rec[i]["polarity"], rec[i]["dt"] = np.nan, "unknown", dt_ns
            rec[i]["flags"], rec[i]["event_length"], rec[i]["time"] = trunc, len(w), (ts * dt_ns * 1000) // 1000
        order = records_sort_order(rec)
        rec = rec[order]
        chunks, cursor = [], 0
        for k, src in enumerate(order):
            w = waves[int(src)][4]
            rec["wave_offset"][k] = cursor
            chunks.append(w.astype(np.uint16))
            cursor += len(w)
        rec["record_id"] = np.arange(len(rec))
        parts.append((rec, np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint16)))
    if not parts:
        return np.zeros(0, dtype=RECORDS_DTYPE), np.zeros(0, dtype=np.uint16)
    return parts[0] if len(parts) == 1 else merge_records_parts(parts)


def vx2730_rows(text: bytes, is_first_file: bool) -> np.ndarray:
    """utils/formats/vx2730.py:165-340: header rows by content (first / second line starting BOARD;CHANNEL;TIMETAG),
    else 2 for the first file of a channel and 0 for the others; blank lines dropped; every other row a list of
    integers.  Columns that are not decimal (the hex FLAGS) become 0 here -- the builder never reads them."""
    lines = text.decode("utf-8", errors="ignore").split("\n")

    def is_header(line):
        f = [x.strip().upper() for x in line.strip().split(";")]
        return len(f) >= 3 and tuple(f[:3]) == ("BOARD", "CHANNEL", "TIMETAG")

    skip = 1 if lines and is_header(lines[0]) else 2 if len(lines) > 1 and lines[1] and is_header(lines[1]) else \
        (2 if is_first_file else 0)
    rows = []
    for line in lines[skip:]:
        line = line.rstrip("\r")
        if not line:
            continue
        vals = []
        for k, f in enumerate(line.split(";")):
            try:
                vals.append(int(f))
            except ValueError:
                if k in (0, 1, 2) or k >= 7:
                    raise
                vals.append(0)
        rows.append(vals)
    if not rows:
        return np.zeros((0, 0), dtype=np.int64)
    if len({len(r) for r in rows}) != 1:
        raise ValueError("rows with differing field counts")
    return np.array(rows, dtype=np.int64)


def build_records_from_vx2730_texts(texts, default_dt_ns: int = 1, baseline_samples=None, epoch_ns=None):
    """records_builder.py:212-302 + 341-426 for the vx2730 adapter: texts = per-channel lists of file contents.  One
    sorted part per file, parts merged with the (part, row) tie-break == the stable sort of all rows."""
    parts = []
    for files in texts:
        for k, text in enumerate(files):
            raw = vx2730_rows(text, is_first_file=(k == 0))
            if raw.size == 0:
                continue
            n = len(raw)
            rec = np.zeros(n, dtype=RECORDS_DTYPE)
            rec["timestamp"] = raw[:, 2]
            rec["board"], rec["channel"] = raw[:, 0].astype(np.int16), raw[:, 1].astype(np.int16)
            if baseline_samples is None:
                b0, b1 = 7, 47
            elif isinstance(baseline_samples, (tuple, list)):
                b0, b1 = 7 + baseline_samples[0], 7 + baseline_samples[1]
            else:
                b0, b1 = 7, 7 + int(baseline_samples)
            b1 = min(b1, raw.shape[1])
            rec["baseline"] = np.mean(raw[:, b0:b1].astype(float), axis=1) if b1 > b0 else np.nan
            rec["baseline_upstream"] = np.nan
            rec["polarity"] = "unknown"
            rec["dt"] = default_dt_ns
            rec["time"] = rec["timestamp"] // 1000 if epoch_ns is None else np.int64(epoch_ns) + rec["timestamp"] // 1000
            wave = raw[:, 7:]
            rec["event_length"] = wave.shape[1]
            order = records_sort_order(rec)
            rec = rec[order]
            rec["wave_offset"] = np.arange(n, dtype=np.int64) * wave.shape[1]
            rec["record_id"] = np.arange(n)
            parts.append((rec, np.asarray(wave[order], dtype=np.uint16).reshape(-1)))
    return merge_records_parts(parts)

