"""DeviceSession: one wfa_ctx (one GPU, one HIP stream) with a resident pool + records SoA.

Host-side mirror of the reference's RecordsView input contract
(waveform_analysis/core/data/records_view.py:16-56): validates the records table, converts it
to the structure-of-arrays the kernels read, and exposes one method per kernel.  All compute
happens in libwfa_hip.so; nothing here falls back to numpy.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import threading
import weakref

import numpy as np

from . import _lib
from .dtypes import (
    BASIC_FEATURES_DTYPE,
    HIT_DTYPE,
    WAVEFORM_WIDTH_DTYPE,
    THRESHOLD_HIT_DTYPE,
    WAVEFORM_WIDTH_INTEGRAL_DTYPE,
)
from .sg_plan import SgPlan, build_plan

REQUIRED_RECORD_FIELDS = ("record_id", "wave_offset", "event_length", "timestamp", "baseline")


def _ptr(arr: np.ndarray | None):
    return None if arr is None else arr.ctypes.data_as(C.c_void_p)


def _col(records: np.ndarray, name: str, dtype, default=None) -> np.ndarray:
    names = records.dtype.names or ()
    if name in names:
        return np.ascontiguousarray(records[name], dtype=dtype)
    if default is None:
        raise ValueError(f"records missing required fields: ['{name}']")
    return np.full(len(records), default, dtype=dtype)


def polarity_codes(records: np.ndarray) -> np.ndarray:
    """'positive'/'negative'/anything else -> WFA_POL_* (records["polarity"], dtypes.py:87)."""
    out = np.zeros(len(records), dtype=np.int8)
    if "polarity" in (records.dtype.names or ()):
        pol = np.asarray(records["polarity"]).astype("U16")
        out[pol == "negative"] = _lib.POL_NEGATIVE
        out[pol == "positive"] = _lib.POL_POSITIVE
    return out


def device_count() -> int:
    n = C.c_int(0)
    _lib.check(_lib.load().wfa_device_count(C.byref(n)))
    return int(n.value)


class DeviceSession:
    """Owns a wfa_ctx.  Not thread-safe: use one session per thread (see device_pool)."""

    def __init__(self, device_id: int = 0):
        self._lib = _lib.load()
        handle = C.c_void_p()
        _lib.check(self._lib.wfa_ctx_create(int(device_id), C.byref(handle)))
        self._h = handle
        self.device_id = int(device_id)
        self.n_records = 0
        self.n_samples = 0
        self.max_len = 0
        self._plan: SgPlan | None = None
        self._keep: list = []  # host arrays referenced by in-flight calls
        # residency: the host array OBJECTS (strong references, compared with `is`) whose contents the device
        # pools hold.  An address / size / dtype key does not identify contents: a freed temporary's address is
        # handed to the next same-shaped array.  Every call that changes a device pool resets these.
        self._res_pool: np.ndarray | None = None
        self._res_filtered: np.ndarray | None = None
        self.uploads = 0  # pool uploads performed (tests, measurement)

    # -- lifetime -------------------------------------------------------------------------------
    def close(self) -> None:
        self.forget_resident()
        if getattr(self, "_h", None):
            self._lib.wfa_ctx_destroy(self._h)
            self._h = None

    def forget_resident(self) -> None:
        """The device pools no longer mirror any host array the session knows of."""
        self._res_pool = None
        self._res_filtered = None

    def _drop_f32_tags(self) -> None:
        """The device float32 buffer is about to be overwritten: forget every host array remembered as its content (the
        filtered twin, and a float32 array that was uploaded as the pool itself)."""
        self._res_filtered = None
        if self._res_pool is not None and self._res_pool.dtype == np.float32:
            self._res_pool = None

    def ensure_pool(self, wave_pool: np.ndarray, cacheable: bool = True) -> bool:
        """Upload `wave_pool` unless this very array object is what the device pool already holds.

        cacheable=False for temporaries (np.ascontiguousarray / astype copies, dense `wave` fields): they are
        uploaded every time and never remembered.  Returns True when an upload happened."""
        if cacheable and self._res_pool is wave_pool:
            return False
        self.upload_pool(wave_pool)
        if cacheable:
            self._res_pool = wave_pool
        return True

    def ensure_filtered_pool(self, pool_f32: np.ndarray, cacheable: bool = True) -> bool:
        if cacheable and self._res_filtered is pool_f32:
            return False
        self.upload_filtered_pool(pool_f32)
        if cacheable:
            self._res_filtered = pool_f32
        return True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- resident inputs ------------------------------------------------------------------------
    def upload_pool(self, wave_pool: np.ndarray) -> None:
        """wave_pool (uint16) or wave_pool_filtered (float32); read-only memmaps are fine."""
        if not isinstance(wave_pool, np.ndarray) or wave_pool.ndim != 1:
            raise ValueError("wave_pool must be a 1-D numpy array")
        if wave_pool.dtype == np.uint16:
            arr = np.ascontiguousarray(wave_pool)
            _lib.check(self._lib.wfa_upload_pool_u16(self._h, _ptr(arr), arr.size))
        elif wave_pool.dtype == np.float32:
            arr = np.ascontiguousarray(wave_pool)
            _lib.check(self._lib.wfa_upload_pool_f32(self._h, _ptr(arr), arr.size))
        else:
            raise ValueError(f"wave_pool dtype must be uint16 or float32, got {wave_pool.dtype}")
        self.forget_resident()
        self.uploads += 1
        self.n_samples = int(wave_pool.size)

    def upload_filtered_pool(self, pool_f32: np.ndarray) -> None:
        """wave_pool_filtered as the float32 twin of the resident wave_pool (same sample count)."""
        arr = np.ascontiguousarray(pool_f32, dtype=np.float32)
        if self.n_samples and arr.size != self.n_samples:
            raise ValueError(f"wave_pool_filtered has {arr.size} samples, wave_pool has {self.n_samples}")
        self._drop_f32_tags()
        _lib.check(self._lib.wfa_upload_pool_f32(self._h, _ptr(arr), arr.size))
        self.uploads += 1

    packed_records = True  # False: always the column route (tests compare the two)
    _PACKED_FIELDS = (("wave_offset", "<i8", True), ("event_length", "<i4", True), ("baseline", "<f8", True),
                      ("polarity", "U", False), ("timestamp", "<i8", True), ("dt", "<i4", False), ("board", "<i2", False),
                      ("channel", "<i2", False), ("record_id", "<i8", True))

    @classmethod
    def _packed_layout(cls, records: np.ndarray):
        """(field byte offsets, polarity width in characters) when the structured array can go to the device as it is:
        one contiguous block of rows whose fields have the reference's types (core/processing/dtypes.py:80-100); None
        for anything else (views with gaps, other integer widths, byte strings, ...): the column route takes those."""
        dt = records.dtype
        if not (records.flags.c_contiguous and records.ndim == 1 and records.strides == (dt.itemsize,)):
            return None
        offsets, pol_chars = [], 0
        for name, kind, _required in cls._PACKED_FIELDS:
            if name not in dt.fields:
                offsets.append(-1)
                continue
            ftype, foff = dt.fields[name][0], dt.fields[name][1]
            if kind == "U":
                if ftype.kind != "U" or ftype.byteorder not in ("=", "<", "|"):
                    return None
                pol_chars = ftype.itemsize // 4
            elif ftype.str != kind:
                return None
            offsets.append(int(foff))
        return np.asarray(offsets, dtype=np.int32), pol_chars

    def upload_records(self, records: np.ndarray, thresholds: np.ndarray | float = 10.0,
                       polarity: np.ndarray | None = None) -> None:
        """records -> device SoA.  `polarity`: optional int8 WFA_POL_* codes overriding records["polarity"].

        A contiguous structured array with the reference's field types is copied to the device as packed rows and
        unpacked there (wfa_upload_records_packed); anything else goes column by column (wfa_upload_records_soa)."""
        if records.dtype.names is None:
            raise ValueError("records must be a structured array")
        missing = [n for n in REQUIRED_RECORD_FIELDS if n not in records.dtype.names]
        if missing:
            raise ValueError(f"records missing required fields: {missing}")
        n = len(records)
        layout = self._packed_layout(records) if self.packed_records else None
        if layout is not None:
            offsets, pol_chars = layout
            thr = np.asarray(thresholds, dtype=np.float64)
            thr_arr = None if thr.ndim == 0 else np.ascontiguousarray(np.broadcast_to(thr, (n,)))
            pol_arr = None if polarity is None else np.ascontiguousarray(polarity, dtype=np.int8)
            if pol_arr is not None and pol_arr.shape != (n,):
                raise ValueError("polarity must hold one code per record")
            max_len, increasing = C.c_int32(0), C.c_int(1)
            _lib.check(self._lib.wfa_upload_records_packed(
                self._h, _ptr(records), n, records.dtype.itemsize, _ptr(offsets), int(pol_chars),
                float(thr) if thr.ndim == 0 else 0.0, _ptr(thr_arr), _ptr(pol_arr), C.byref(max_len), C.byref(increasing)))
            if not increasing.value:  # rare: ids out of order -- uniqueness the way the column route checks it
                self._check_unique_ids(np.ascontiguousarray(records["record_id"], dtype=np.int64))
            self.n_records = n
            self.max_len = int(max_len.value)
            return
        rid = _col(records, "record_id", np.int64)
        self._check_unique_ids(rid)
        thr = np.ascontiguousarray(np.broadcast_to(np.asarray(thresholds, dtype=np.float64), (n,)))
        cols = [
            _col(records, "wave_offset", np.int64),
            _col(records, "event_length", np.int32),
            _col(records, "baseline", np.float64),
            polarity_codes(records) if polarity is None else np.ascontiguousarray(polarity, dtype=np.int8),
            thr,
            _col(records, "timestamp", np.int64),
            _col(records, "dt", np.int32, default=1),
            _col(records, "board", np.int16, default=0),
            _col(records, "channel", np.int16, default=0),
            rid,
        ]
        _lib.check(self._lib.wfa_upload_records_soa(self._h, n, *[_ptr(c) for c in cols]))
        self.n_records = n
        self.max_len = int(cols[1].max()) if n else 0

    @staticmethod
    def _check_unique_ids(rid: np.ndarray) -> None:
        n = len(rid)
        if n and len(np.unique(rid)) != n:
            ordered = np.sort(rid)
            dup = ordered[np.flatnonzero(np.diff(ordered) == 0)[0]]
            raise ValueError(f"records field record_id must be unique, got duplicate {int(dup)}")

    def set_sg_plan(self, sg_window_size: int = 11, sg_poly_order: int = 2) -> SgPlan:
        plan = build_plan(int(sg_window_size), int(sg_poly_order))
        _lib.check(
            self._lib.wfa_set_sg_plan(
                self._h, plan.window, plan.polyorder, _ptr(plan.tab), _ptr(plan.symmetric),
                int(plan.int_ok), _ptr(plan.itab), plan.den, plan.den_edge, plan.guard, plan.guard_edge,
            )
        )
        self._plan = plan
        return plan

    # -- kernels --------------------------------------------------------------------------------
    def baseline_mean(self, start: int, end: int, update_records: bool = False) -> np.ndarray:
        out = np.empty(self.n_records, dtype=np.float64)
        _lib.check(self._lib.wfa_baseline_mean(self._h, int(start), int(end), int(update_records), _ptr(out)))
        return out

    def filter_keep_output(self, keep: bool) -> None:
        """keep=True: the next savgol/sosfiltfilt calls write only their records' slices into the output."""
        _lib.check(self._lib.wfa_filter_keep_output(self._h, int(bool(keep))))

    def download_filtered(self) -> np.ndarray:
        """The resident float32 pool (after one or more filter calls)."""
        out = np.empty(self.n_samples, dtype=np.float32)
        _lib.check(self._lib.wfa_download_pool_f32(self._h, _ptr(out), out.size))
        return out

    def savgol(self, download: bool = True) -> np.ndarray | None:
        out = np.empty(self.n_samples, dtype=np.float32) if download else None
        self._drop_f32_tags()  # the device float32 pool is this filter's output now
        _lib.check(self._lib.wfa_savgol(self._h, _ptr(out)))
        return out

    def sosfiltfilt(self, sos: np.ndarray, zi: np.ndarray, padlen: int, download: bool = True) -> np.ndarray | None:
        """Butterworth branch of wave_pool_filtered: sos (n x 6) from scipy.signal.butter(output="sos"),
        zi (n x 2) from scipy.signal.sosfilt_zi, padlen per filtering.py:198-203."""
        sos = np.ascontiguousarray(sos, dtype=np.float64)
        zi = np.ascontiguousarray(zi, dtype=np.float64)
        if sos.ndim != 2 or sos.shape[1] != 6 or zi.shape != (sos.shape[0], 2):
            raise ValueError("sos must be (n_sections, 6) and zi (n_sections, 2)")
        out = np.empty(self.n_samples, dtype=np.float32) if download else None
        self._drop_f32_tags()
        _lib.check(self._lib.wfa_sosfiltfilt(self._h, int(sos.shape[0]), _ptr(sos), _ptr(zi), int(padlen), _ptr(out)))
        return out

    def _fill_hits(self, n: int) -> np.ndarray:
        out = np.empty(n, dtype=THRESHOLD_HIT_DTYPE)
        _lib.check(self._lib.wfa_threshold_hits_fill(self._h, _ptr(out), n))
        return out

    def threshold_hits(self, source: int = _lib.SRC_RAW, left_extension: int = 2, right_extension: int = 2,
                       max_len: int = 0, download: bool = True) -> np.ndarray | int:
        n = C.c_int64(0)
        _lib.check(self._lib.wfa_threshold_hits_count(self._h, int(source), int(left_extension),
                                                      int(right_extension), int(max_len), C.byref(n)))
        return self._fill_hits(int(n.value)) if download else int(n.value)

    def fused_baseline_filter_hits(self, baseline_window: tuple[int, int] = (0, 0), left_extension: int = 2,
                                   right_extension: int = 2, max_len: int = 0,
                                   download: bool = True) -> np.ndarray | int:
        n = C.c_int64(0)
        _lib.check(self._lib.wfa_fused_baseline_filter_hits(
            self._h, int(baseline_window[0]), int(baseline_window[1]), int(left_extension),
            int(right_extension), int(max_len), C.byref(n)))
        return self._fill_hits(int(n.value)) if download else int(n.value)

    def hits_enqueue(self, source: int = _lib.SRC_SG_FUSED, baseline_window: tuple[int, int] = (0, 0),
                     left_extension: int = 2, right_extension: int = 2, max_len: int = 0) -> None:
        """Queue a hit pass without waiting for it (see wfa_hits_enqueue); hits_wait() / _fill_hits deliver."""
        _lib.check(self._lib.wfa_hits_enqueue(self._h, int(source), int(baseline_window[0]), int(baseline_window[1]),
                                              int(left_extension), int(right_extension), int(max_len)))

    def hits_wait(self) -> int:
        n = C.c_int64(0)
        _lib.check(self._lib.wfa_hits_wait(self._h, C.byref(n)))
        return int(n.value)

    def find_peaks(self, source: int = _lib.SRC_F32, use_derivative: bool = True, height: float = 30.0,
                   distance: int = 2, prominence: float = 0.7, width: float = 4, threshold: float | None = None,
                   height_method: str = "minmax", height_window_extension: int = 4,
                   dense_rows: bool | int = False) -> np.ndarray:
        """find_peaks-based hit detector (HitFinderPlugin) -> HIT_DTYPE rows.  dense_rows: True / 1 = the dense branch
        (WFA_PEAK_SIGNAL_ROWS), the uploaded records describe the rows of an st_waveforms / filtered_waveforms array;
        2 = the streaming detector's float64 rows (WFA_PEAK_SIGNAL_ROWS_F64)."""
        if height_method not in ("minmax", "diff"):
            raise ValueError(f"不支持的峰高计算方法: {height_method}")  # peak_finding.py:612
        n = C.c_int64(0)
        _lib.check(self._lib.wfa_find_peaks_count(
            self._h, int(source), int(dense_rows), int(bool(use_derivative)), float(height),
            int(threshold is not None),
            float(threshold or 0.0), int(distance), float(prominence), float(width),
            1 if height_method == "diff" else 0, int(height_window_extension), C.byref(n)))
        out = np.empty(int(n.value), dtype=HIT_DTYPE)
        _lib.check(self._lib.wfa_find_peaks_fill(self._h, _ptr(out), int(n.value)))
        return out

    def find_hits_legacy(self, source: int, n_rows: int, row_length: int, baselines: np.ndarray,
                         threshold: float) -> tuple[np.ndarray, np.ndarray]:
        """(event_index, start sample) of every (baseline - wave) > threshold run on the resident dense matrix."""
        b = np.ascontiguousarray(baselines, dtype=np.float64)
        if b.shape != (int(n_rows),):
            raise ValueError("baselines must hold one value per row")
        n = C.c_int64(0)
        _lib.check(self._lib.wfa_find_hits_count(self._h, int(source), int(n_rows), int(row_length), _ptr(b),
                                                 float(threshold), C.byref(n)))
        ev = np.empty(int(n.value), dtype=np.int64)
        t = np.empty(int(n.value), dtype=np.int64)
        _lib.check(self._lib.wfa_find_hits_fill(self._h, int(n.value), _ptr(ev), _ptr(t)))
        return ev, t

    def waveform_width(self, source: int, position: np.ndarray, row_index: np.ndarray, n_rows: int, row_length: int,
                       rise_low: float = 0.1, rise_high: float = 0.9, fall_high: float = 0.9, fall_low: float = 0.1,
                       sampling_rate: float = 0.5, interpolation: bool = True) -> tuple[np.ndarray, np.ndarray]:
        """Rise/fall/total width per hit on the resident dense wave matrix (WaveformWidthPlugin).

        Returns (rows, valid): WAVEFORM_WIDTH_DTYPE rows for every hit (id fields zero) and the bool mask of
        hits the reference keeps."""
        pos = np.ascontiguousarray(position, dtype=np.int64)
        row = np.ascontiguousarray(row_index, dtype=np.int64)
        if pos.shape != row.shape or pos.ndim != 1:
            raise ValueError("position and row_index must be 1-D arrays of the same length")
        out = np.zeros(len(pos), dtype=WAVEFORM_WIDTH_DTYPE)
        valid = np.zeros(len(pos), dtype=np.uint8)
        _lib.check(self._lib.wfa_waveform_width(
            self._h, int(source), len(pos), _ptr(pos), _ptr(row), int(n_rows), int(row_length), float(rise_low),
            float(rise_high), float(fall_high), float(fall_low), float(sampling_rate), int(bool(interpolation)),
            _ptr(out), _ptr(valid)))
        return out, valid.astype(bool)

    # ---- records builder -----------------------------------------------------------------------------------------
    def records_sort_order(self, timestamp, pid, board, channel) -> np.ndarray:
        """np.lexsort((seq, channel, board, pid, timestamp)) on the device (records_builder.py:115-120)."""
        ts = np.ascontiguousarray(timestamp, dtype=np.int64)
        cols = [ts, np.ascontiguousarray(pid, dtype=np.int32), np.ascontiguousarray(board, dtype=np.int16),
                np.ascontiguousarray(channel, dtype=np.int16)]
        if any(c.shape != ts.shape or c.ndim != 1 for c in cols):
            raise ValueError("sort columns must be 1-D arrays of one length")
        order = np.empty(len(ts), dtype=np.int64)
        _lib.check(self._lib.wfa_records_sort(self._h, len(ts), *[_ptr(c) for c in cols], _ptr(order)))
        return order

    def csv_decode(self, text, delimiter: str = ";", samples_start: int = 7, meta_cols=(0, 1, 2),
                   download_samples: bool = False) -> dict:
        """Delimiter-separated integer text (CAEN VX2730 CSV rows, header rows already removed) -> per-row tables.

        Returns meta (n_rows x len(meta_cols) int64), row_offset, n_fields, sample_offset, n_samples and, on request,
        the ragged uint16 samples; the samples stay on the device as the source of pool_gather(src_pool=None)."""
        buf = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else \
            np.ascontiguousarray(text, dtype=np.uint8)
        n_rows, n_samples = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.wfa_csv_decode_count(self._h, _ptr(buf), buf.size, ord(delimiter), int(samples_start),
                                                  C.byref(n_rows), C.byref(n_samples)))
        n, ns = int(n_rows.value), int(n_samples.value)
        cols = np.ascontiguousarray(meta_cols, dtype=np.int32)
        out = {"meta": np.zeros((n, len(cols)), dtype=np.int64), "row_offset": np.zeros(n, dtype=np.int64),
               "n_fields": np.zeros(n, dtype=np.int32), "sample_offset": np.zeros(n, dtype=np.int64), "n_samples": ns,
               "samples": np.zeros(ns, dtype=np.uint16) if download_samples else None}
        _lib.check(self._lib.wfa_csv_decode_fill(self._h, n, len(cols), _ptr(cols), _ptr(out["meta"]),
                                                 _ptr(out["row_offset"]), _ptr(out["n_fields"]),
                                                 _ptr(out["sample_offset"]), _ptr(out["samples"]), ns))
        return out

    def pool_gather(self, src_offset, length, src_pool: np.ndarray | None, download: bool = True,
                    src_samples: int = 0):
        """Pack wave slices (given in output order) into the resident wave_pool -> (out_offset, pool | None).
        src_pool=None with src_samples=n: the source is the samples the last csv_decode left on the device."""
        so = np.ascontiguousarray(src_offset, dtype=np.int64)
        ln = np.ascontiguousarray(length, dtype=np.int32)
        total = int(np.maximum(ln, 0).astype(np.int64).sum())
        out_off = np.empty(len(so), dtype=np.int64)
        out = np.empty(total, dtype=np.uint16) if download else None
        self.forget_resident()  # the resident wave_pool becomes the packed output
        if src_pool is None:
            _lib.check(self._lib.wfa_pool_gather(self._h, len(so), _ptr(so), _ptr(ln), None, int(src_samples),
                                                 _ptr(out_off), _ptr(out), total))
            self.n_samples = total
            self.n_records = 0
            return out_off, out
        src = np.ascontiguousarray(src_pool)
        if src.dtype == np.int16:
            src = src.view(np.uint16)  # _clip_wave_to_uint16: astype(uint16) keeps the bit pattern
        if src.dtype != np.uint16 or src.ndim != 1:
            raise ValueError("src_pool must be a flat uint16 / int16 array")
        _lib.check(self._lib.wfa_pool_gather(self._h, len(so), _ptr(so), _ptr(ln), _ptr(src), src.size, _ptr(out_off),
                                             _ptr(out), total))
        self.n_samples = total
        self.n_records = 0
        return out_off, out

    # ---- hit-table stages (device sort + scans) ------------------------------------------------------------
    @staticmethod
    def _hit_cols(timestamp, position, start, end, dt, board, channel, record_id):
        cols = [np.ascontiguousarray(timestamp, dtype=np.int64), np.ascontiguousarray(position, dtype=np.int64),
                np.ascontiguousarray(start, dtype=np.int32), np.ascontiguousarray(end, dtype=np.int32),
                np.ascontiguousarray(dt, dtype=np.int32), np.ascontiguousarray(board, dtype=np.int16),
                np.ascontiguousarray(channel, dtype=np.int16), np.ascontiguousarray(record_id, dtype=np.int64)]
        n = len(cols[0])
        if any(c.ndim != 1 or len(c) != n for c in cols):
            raise ValueError("hit columns must be 1-D arrays of one length")
        return n, cols

    def hit_merge_clusters(self, timestamp, position, edge_start, edge_end, dt, board, channel,
                           merge_gap_ns: float, max_total_width_ns: float) -> tuple[np.ndarray, np.ndarray]:
        """Per-channel chain merge (hit_merge.py:115-181): (order, cluster_offset)."""
        n, cols = self._hit_cols(timestamp, position, edge_start, edge_end, dt, board, channel, timestamp)
        m = C.c_int64(0)
        _lib.check(self._lib.wfa_hit_merge_count(self._h, n, *[_ptr(c) for c in cols[:7]], float(merge_gap_ns),
                                                 float(max_total_width_ns), C.byref(m)))
        order = np.empty(n, np.int64)
        offset = np.empty(int(m.value) + 1, np.int64)
        _lib.check(self._lib.wfa_hit_merge_fill(self._h, n, int(m.value), _ptr(order), _ptr(offset)))
        return order, offset

    def hit_merge_emit(self, timestamp, sample_start, sample_end, record_id, height, integral, member_hit,
                       cluster_offset) -> dict:
        """Per-cluster anchor / height / integral / sample window / width (hit_merge.py:256-322)."""
        ts = np.ascontiguousarray(timestamp, dtype=np.int64)
        s = np.ascontiguousarray(sample_start, dtype=np.int32)
        e = np.ascontiguousarray(sample_end, dtype=np.int32)
        rid = np.ascontiguousarray(record_id, dtype=np.int64)
        h = np.ascontiguousarray(height, dtype=np.float32)
        q = np.ascontiguousarray(integral, dtype=np.float32)
        mem = np.ascontiguousarray(member_hit, dtype=np.int64)
        off = np.ascontiguousarray(cluster_offset, dtype=np.int64)
        n, k = len(ts), len(off) - 1
        out = {"anchor": np.empty(k, np.int64), "height": np.empty(k, np.float32), "integral": np.empty(k, np.float32),
               "sample_start": np.empty(k, np.int32), "sample_end": np.empty(k, np.int32), "width": np.empty(k, np.float32)}
        _lib.check(self._lib.wfa_hit_merge_emit(self._h, n, _ptr(ts), _ptr(s), _ptr(e), _ptr(rid), _ptr(h), _ptr(q),
                                                len(mem), _ptr(mem), k, _ptr(off),
                                                *[_ptr(out[f]) for f in ("anchor", "height", "integral", "sample_start",
                                                                        "sample_end", "width")]))
        return out

    def group_hit_windows(self, timestamp, position, sample_start, sample_end, dt, board, channel, record_id,
                          time_window_ns: float, abs_start_fix=None, abs_end_fix=None) -> dict:
        """Gap-chained event grouping (event_grouping.py:286-471): order, event_start, t_min, t_max."""
        n, cols = self._hit_cols(timestamp, position, sample_start, sample_end, dt, board, channel, record_id)
        f0 = None if abs_start_fix is None else np.ascontiguousarray(abs_start_fix, dtype=np.float64)
        f1 = None if abs_end_fix is None else np.ascontiguousarray(abs_end_fix, dtype=np.float64)
        m = C.c_int64(0)
        _lib.check(self._lib.wfa_group_hit_windows_count(self._h, n, *[_ptr(c) for c in cols], _ptr(f0), _ptr(f1),
                                                         float(time_window_ns), C.byref(m)))
        k = int(m.value)
        out = {"order": np.empty(n, np.int64), "event_start": np.empty(k + 1, np.int64),
               "t_min": np.empty(k, np.int64), "t_max": np.empty(k, np.int64)}
        _lib.check(self._lib.wfa_group_hit_windows_fill(self._h, n, k, _ptr(out["order"]), _ptr(out["event_start"]),
                                                        _ptr(out["t_min"]), _ptr(out["t_max"])))
        return out

    def group_multi_channel(self, timestamp, channel, time_window_ps: float) -> tuple[np.ndarray, np.ndarray]:
        """Legacy fixed-window grouping (group_multi_channel_hits): (order, bounds) -- input rows event-major, by channel
        inside an event; event e = order[bounds[e]:bounds[e + 1]]."""
        ts = np.ascontiguousarray(timestamp, dtype=np.int64)
        ch = np.ascontiguousarray(channel, dtype=np.int64)
        if ts.shape != ch.shape or ts.ndim != 1:
            raise ValueError("timestamp and channel must be one-dimensional and of equal length")
        n = int(ts.size)
        m = C.c_int64(0)
        _lib.check(self._lib.wfa_group_multi_channel_count(self._h, n, _ptr(ts), _ptr(ch), float(time_window_ps), C.byref(m)))
        k = int(m.value)
        order, bounds = np.empty(n, np.int64), np.empty(k + 1, np.int64)
        _lib.check(self._lib.wfa_group_multi_channel_fill(self._h, n, k, _ptr(order), _ptr(bounds)))
        return order, bounds

    # ---- the same stages on device-resident THRESHOLD_HIT_DTYPE rows (no host columns in, none needed out) ------------
    def hit_rows_source(self, which: str = "hits") -> None:
        """Rows the *_resident stages read: "hits" = the last hit pass of this session, "gather" = the rows the last
        rccl_gather_rows left on the root."""
        _lib.check(self._lib.wfa_hit_rows_source(self._h, {"hits": 1, "gather": 2}[which]))

    def group_hit_windows_resident(self, n: int, time_window_ns: float) -> dict:
        """group_hit_windows on the n resident rows (sample window = edge_start / edge_end, as for hit_threshold rows)."""
        m = C.c_int64(0)
        _lib.check(self._lib.wfa_group_hit_windows_count(self._h, int(n), *([None] * 10), float(time_window_ns), C.byref(m)))
        k = int(m.value)
        out = {"order": np.empty(n, np.int64), "event_start": np.empty(k + 1, np.int64),
               "t_min": np.empty(k, np.int64), "t_max": np.empty(k, np.int64)}
        _lib.check(self._lib.wfa_group_hit_windows_fill(self._h, int(n), k, _ptr(out["order"]), _ptr(out["event_start"]),
                                                        _ptr(out["t_min"]), _ptr(out["t_max"])))
        return out

    def hit_merge_clusters_resident(self, n: int, merge_gap_ns: float, max_total_width_ns: float):
        """hit_merge_clusters on the n resident rows: (order, cluster_offset)."""
        m = C.c_int64(0)
        _lib.check(self._lib.wfa_hit_merge_count(self._h, int(n), *([None] * 7), float(merge_gap_ns),
                                                 float(max_total_width_ns), C.byref(m)))
        order = np.empty(n, np.int64)
        offset = np.empty(int(m.value) + 1, np.int64)
        _lib.check(self._lib.wfa_hit_merge_fill(self._h, int(n), int(m.value), _ptr(order), _ptr(offset)))
        return order, offset

    def basic_features(self, source: int = _lib.SRC_RAW, height_range=(40, 90), area_range=(0, None),
                       fixed_baseline: np.ndarray | None = None) -> np.ndarray:
        out = np.zeros(self.n_records, dtype=BASIC_FEATURES_DTYPE)
        h0, h1 = height_range
        a0, a1 = area_range
        fb = None if fixed_baseline is None else np.ascontiguousarray(fixed_baseline, dtype=np.float64)
        if fb is not None and len(fb) != self.n_records:
            raise ValueError("fixed_baseline must have one entry per record")
        _lib.check(self._lib.wfa_basic_features(
            self._h, int(source), int(h0 or 0), int(h1 or 0), int(h1 is not None),
            int(a0 or 0), int(a1 or 0), int(a1 is not None), _ptr(fb), _ptr(out)))
        return out

    def width_integral(self, source: int = _lib.SRC_RAW, q_low: float = 0.1, q_high: float = 0.9,
                       dt: float = 2.0) -> np.ndarray:
        out = np.zeros(self.n_records, dtype=WAVEFORM_WIDTH_INTEGRAL_DTYPE)
        _lib.check(self._lib.wfa_width_integral(self._h, int(source), float(q_low), float(q_high),
                                                float(dt), _ptr(out)))
        return out

    def features_both(self, height_range=(40, 90), area_range=(0, None), q_low: float = 0.1, q_high: float = 0.9,
                      dt: float = 2.0, download: bool = True):
        """basic_features and width_integral rows of the raw pool from one read of it (wfa_features_both) ->
        (BASIC_FEATURES_DTYPE rows, WAVEFORM_WIDTH_INTEGRAL_DTYPE rows), or (None, None) with download=False."""
        h0, h1 = height_range
        a0, a1 = area_range
        ob = np.zeros(self.n_records, dtype=BASIC_FEATURES_DTYPE) if download else None
        ow = np.zeros(self.n_records, dtype=WAVEFORM_WIDTH_INTEGRAL_DTYPE) if download else None
        _lib.check(self._lib.wfa_features_both(
            self._h, int(h0 or 0), int(h1 or 0), int(h1 is not None), int(a0 or 0), int(a1 or 0), int(a1 is not None),
            float(q_low), float(q_high), float(dt), _ptr(ob), _ptr(ow)))
        return ob, ow

    def sync(self) -> None:
        _lib.check(self._lib.wfa_sync(self._h))

    def release_scratch(self) -> int:
        """Give back the device scratch later calls rebuild by themselves (wfa_release_scratch); returns the bytes freed.
        The resident pools, records, plan and the rows of the last passes stay."""
        freed = C.c_int64(0)
        _lib.check(self._lib.wfa_release_scratch(self._h, C.byref(freed)))
        return int(freed.value)

    def last_h2d_rate(self) -> float:
        """GB/s of the last large upload through the pinned staging ring."""
        v = C.c_double(0.0)
        _lib.check(self._lib.wfa_last_h2d_rate(self._h, C.byref(v)))
        return float(v.value)

    def set_option(self, name: str, value: bool | int = True) -> None:
        """Select between code paths with identical results (wfa_set_option: tests and measurement only)."""
        _lib.check(self._lib.wfa_set_option(self._h, name.encode(), int(value)))

    # -- measurement ----------------------------------------------------------------------------
    def profile(self, on: bool = True) -> None:
        _lib.check(self._lib.wfa_profile_enable(self._h, int(on)))
        _lib.check(self._lib.wfa_profile_reset(self._h))

    def profile_report(self) -> dict[str, tuple[float, int]]:
        out = {}
        name = C.create_string_buffer(128)
        ms, cnt = C.c_double(0), C.c_int64(0)
        idx = 0
        while self._lib.wfa_profile_get(self._h, idx, name, len(name), C.byref(ms), C.byref(cnt)) == 0:
            out[name.value.decode()] = (float(ms.value), int(cnt.value))
            idx += 1
        return out

    # -- RCCL -----------------------------------------------------------------------------------
    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().wfa_rccl_unique_id(buf))
        return buf.raw

    def rccl_init(self, rank: int, n_ranks: int, unique_id: bytes) -> None:
        buf = C.create_string_buffer(unique_id, 128)
        _lib.check(self._lib.wfa_rccl_init(self._h, int(rank), int(n_ranks), buf))
        self.rank, self.n_ranks = int(rank), int(n_ranks)

    def rccl_gather_append(self, on: bool = True) -> None:
        """on: the following gathers append their rows on the root behind those already there (one exchange per shard
        or time-range chunk, one table for event grouping); off: one table per gather again, the table is dropped."""
        _lib.check(self._lib.wfa_rccl_gather_append(self._h, int(bool(on))))

    def rccl_gather_rows(self, rows: np.ndarray | None, n_rows: int, row_dtype: np.dtype, root: int = 0,
                         download: bool = True):
        """Gather structured rows from all ranks to `root` (rank order).  rows=None sends the
        device-resident hit rows of the last hit pass.  Returns (counts, rows-or-None).  download=False leaves the
        gathered rows on the root's device only (hit_rows_source("gather") + the *_resident stages read them)."""
        row_dtype = np.dtype(row_dtype)
        counts = np.zeros(self.n_ranks, dtype=np.int64)
        _lib.check(self._lib.wfa_rccl_allgather_counts(self._h, int(n_rows), _ptr(counts)))
        out = np.empty(int(counts.sum()), dtype=row_dtype) if (self.rank == root and download) else None
        src = None if rows is None else np.ascontiguousarray(rows)
        _lib.check(self._lib.wfa_rccl_gather_rows(self._h, _ptr(src), int(n_rows), row_dtype.itemsize,
                                                  int(root), _ptr(counts), _ptr(out)))
        return counts, out


# ---- device pool for the streaming dispatcher ----------------------------------------------------
class DevicePool:
    """Sessions for worker threads: chunk k runs on GPU (k mod n_devices) on its own HIP stream.

    The reference dispatches chunks to an ExecutorManager thread pool
    (waveform_analysis/core/plugins/core/streaming.py:740-860).  Two ways to get a session:

    * `borrow()` -- context manager for short-lived workers (the streaming drivers): takes a session from a free
      list, creates one while fewer than `max_sessions` are alive, otherwise waits for one to come back.  Executors
      come and go; the sessions (and their device buffers, stream, events) are reused, so repeated `compute()`
      calls hold at most `max_sessions` contexts.
    * `session()` -- the session bound to the calling thread (static plugins on the Context's thread).  The pool keeps
      only weak references to these: when the thread goes away its session is closed.
    """

    def __init__(self, device_ids: list[int] | None = None, max_sessions: int | None = None,
                 session_factory=None):
        if device_ids is not None:
            ids = list(device_ids)
        else:
            ids = list(range(max(device_count(), 1))) if session_factory is None else [0]
        if not ids:
            raise _lib.WfaError(_lib.WFA_E_HIP, "no HIP device visible")
        self.device_ids = ids
        self.max_sessions = int(max_sessions) if max_sessions else max(8, 2 * len(ids))
        self._factory = session_factory or DeviceSession
        self._local = threading.local()
        self._lock = threading.Condition()
        self._next = 0
        self._free: list[DeviceSession] = []
        self._borrowable = 0                       # sessions created for borrow() and still alive
        self._thread_bound = weakref.WeakSet()     # sessions handed out by session()

    def _new_session(self) -> DeviceSession:
        dev = self.device_ids[self._next % len(self.device_ids)]
        self._next += 1
        return self._factory(dev)

    def session(self) -> DeviceSession:
        s = getattr(self._local, "session", None)
        if s is None:
            with self._lock:
                s = self._new_session()
                self._thread_bound.add(s)
            self._local.session = s
        return s

    def peek_session(self) -> DeviceSession | None:
        """The session bound to the calling thread, if it has one (never creates one)."""
        return getattr(self._local, "session", None)

    def drop_session(self) -> bool:
        """Close the calling thread's session (a plugin failed on it: its device state is not trusted any more); the
        next session() call of this thread starts from a new context."""
        s = getattr(self._local, "session", None)
        if s is None:
            return False
        with self._lock:
            self._thread_bound.discard(s)
        self._local.session = None
        s.close()
        return True

    @contextlib.contextmanager
    def borrow(self):
        with self.borrow_many(1) as got:
            yield got[0]

    @contextlib.contextmanager
    def borrow_many(self, n: int):
        """`n` sessions at once, taken atomically (two callers that each hold one session and wait for a second would
        wait for ever), on as many different devices as the pool has: free sessions are preferred by device, new ones
        are created round-robin.  Raises when the pool can never hold that many."""
        n = int(n)
        if n < 1 or n > self.max_sessions:
            raise ValueError(f"cannot borrow {n} sessions from a pool of max_sessions={self.max_sessions}")
        with self._lock:
            while len(self._free) + (self.max_sessions - self._borrowable) < n:
                self._lock.wait()
            got: list[DeviceSession] = []
            for _ in range(n):
                used = [getattr(s, "device_id", None) for s in got]
                # a free session on the device this borrower uses least, else a new one (round-robin over the devices)
                pick = min(self._free, key=lambda s: used.count(getattr(s, "device_id", None)), default=None)
                if pick is not None and (used.count(getattr(pick, "device_id", None)) == 0
                                         or self._borrowable >= self.max_sessions):
                    self._free.remove(pick)
                    got.append(pick)
                elif self._borrowable < self.max_sessions:
                    got.append(self._new_session())
                    self._borrowable += 1
                else:
                    self._free.remove(pick)
                    got.append(pick)
        try:
            yield got
        finally:
            with self._lock:
                self._free.extend(got)
                self._lock.notify_all()

    @property
    def live_sessions(self) -> int:
        with self._lock:
            return self._borrowable + len(self._thread_bound)

    def close(self) -> None:
        with self._lock:
            for s in self._free:
                s.close()
            self._borrowable -= len(self._free)
            self._free.clear()
            for s in list(self._thread_bound):
                s.close()
            self._thread_bound.clear()
        self._local = threading.local()


_default_pool: DevicePool | None = None
_default_lock = threading.Lock()


def default_pool() -> DevicePool:
    global _default_pool
    with _default_lock:
        if _default_pool is None:
            _default_pool = DevicePool()
        return _default_pool


__all__ = ["DeviceSession", "DevicePool", "default_pool", "device_count", "polarity_codes"]
