"""Dense (st_waveforms / filtered_waveforms) inputs as a records + pool view.

The reference's dense branches (`wave_source` = st_waveforms | filtered_waveforms | auto) read a structured
array whose `wave` field is an (n_events, n_samples) int16 (ST_WAVEFORM_DTYPE, processing/dtypes.py:18-64) or
float32 (filtered_waveforms, cpu/filtering.py:133-158) matrix and always use the WHOLE row.  The kernels read
a flat pool with per-record (offset, length): row i becomes the record (i * L, L) of the row-major matrix, so
every dense row is a uniform-length contiguous record and takes the same kernels as the records path.
"""

from __future__ import annotations

import numpy as np

from . import _lib

DENSE_RECORD_DTYPE = np.dtype(
    [
        ("timestamp", "i8"),
        ("board", "i2"),
        ("channel", "i2"),
        ("baseline", "f8"),
        ("polarity", "U8"),
        ("record_id", "i8"),
        ("dt", "i4"),
        ("wave_offset", "i8"),
        ("event_length", "i4"),
    ]
)


def matrix_pool(wave: np.ndarray, what: str = "st_waveforms") -> tuple[np.ndarray, int, int]:
    """(flat pool, source code, row length) of an (n_events, n_samples) int16 / uint16 / float32 matrix.

    int16 rows are viewed as uint16: ADC codes are non-negative (14/16-bit unsigned converters); a negative
    sample is refused rather than reinterpreted."""
    if wave.ndim != 2:
        raise ValueError(f"{what}['wave'] must be 2D (n_events, n_samples)")
    L = int(wave.shape[1])
    if wave.dtype == np.float32:
        return np.ascontiguousarray(wave).reshape(-1), _lib.SRC_F32, L
    if wave.dtype in (np.int16, np.uint16):
        flat = np.ascontiguousarray(wave).reshape(-1)
        if wave.dtype == np.int16:
            if flat.size and int(flat.min()) < 0:
                raise ValueError(f"{what}['wave'] holds negative samples; the HIP backend reads unsigned ADC codes")
            flat = flat.view(np.uint16)
        return flat, _lib.SRC_RAW, L
    raise ValueError(f"{what}['wave'] must be int16 or float32, got {wave.dtype}")


def dense_pool(waveform_data: np.ndarray, what: str = "st_waveforms") -> tuple[np.ndarray, int, int]:
    """matrix_pool of a dense structured array's `wave` field."""
    names = waveform_data.dtype.names or ()
    if "wave" not in names:
        raise ValueError(f"{what} missing required 'wave' field")
    return matrix_pool(waveform_data["wave"], what)


def dense_records(waveform_data: np.ndarray, row_length: int, *, keep_record_id: bool = False,
                  truncate_to_event_length: bool = False) -> np.ndarray:
    """Per-row records for the row-major pool of `dense_pool` (whole rows, as the dense branches read them).

    keep_record_id: carry the array's own record_id (the hit tables report it); otherwise rows are numbered.
    truncate_to_event_length: rows with 0 < event_length < row length are cut there (peak_finding.py:336-343)."""
    n = len(waveform_data)
    names = waveform_data.dtype.names or ()
    rec = np.zeros(n, dtype=DENSE_RECORD_DTYPE)
    for name, default in (("timestamp", 0), ("board", 0), ("channel", 0), ("baseline", np.nan), ("dt", 1)):
        rec[name] = waveform_data[name] if name in names else default
    rec["polarity"] = waveform_data["polarity"] if "polarity" in names else "negative"
    if keep_record_id and "record_id" in names:
        rec["record_id"] = waveform_data["record_id"]
    else:
        rec["record_id"] = np.arange(n, dtype=np.int64)  # kernels address rows by position
    rec["wave_offset"] = np.arange(n, dtype=np.int64) * int(row_length)
    rec["event_length"] = int(row_length)
    if truncate_to_event_length and "event_length" in names:
        ev = np.asarray(waveform_data["event_length"], dtype=np.int64)
        cut = (ev > 0) & (ev < int(row_length))
        rec["event_length"][cut] = ev[cut]
    return rec


def dense_polarity_wave_rule(waveform_data: np.ndarray) -> np.ndarray:
    """Polarity codes of the st branch of BasicFeaturesPlugin (basic_features.py:239-262): formulas stay
    wave-based; only the literal "positive" flips the sign."""
    out = np.zeros(len(waveform_data), dtype=np.int8)
    if "polarity" in (waveform_data.dtype.names or ()):
        pol = np.asarray(waveform_data["polarity"]).astype("U16")
        out[pol == "positive"] = _lib.POL_POSITIVE_WAVE
    return out
