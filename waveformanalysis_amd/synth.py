"""Synthetic V1725 / VX2730-like runs (records + wave_pool) for tests and bench.

Generator specified in SURVEY.md section 8d:
  * per channel pedestal ~ U{7800..8200} (14-bit ADC), white noise round(N(0, 3)),
  * with probability 0.80 one, 0.15 two, 0.05 zero negative pulses per record,
    start ~ U[60, L-120], amplitude 10**U(1.3, 3.3) ADC,
    shape A*(exp(-t/tau_f) - exp(-t/tau_r)), tau_r = 4, tau_f ~ U(10, 60) samples,
  * clip to [0, 16383], cast uint16,
  * timestamps: Poisson process per channel (mean gap 10 us) in ps, records sorted
    with the reference rule (timestamp, pid, board, channel, seq)
    (reference: waveform_analysis/core/processing/records_builder.py:115-120),
  * baseline = mean of the first 40 samples (records_builder.py:243-257,
    utils/formats/vx2730.py:90-93), polarity = "unknown".

The generator is deterministic per (seed, chunk index) so any slice of a large
run can be regenerated for a CPU check without materialising the whole run.
"""

from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
import os

import numpy as np

from .dtypes import RECORDS_DTYPE

SEED_BASE = 20260424
BASELINE_SAMPLES = 40

PRESETS = {
    # name: (L, dt_ns, n_boards, channels_per_board)
    "v1725": (800, 4, 1, 16),
    "v1725_1ch": (800, 4, 1, 1),
    "vx2730": (1500, 2, 8, 32),
}


def _records_meta(n_records: int, L: int, dt_ns: int, n_boards: int, ch_per_board: int, seed: int):
    """Timestamps / channel assignment / sort, without waves."""
    rng = np.random.default_rng(seed)
    n_ch = n_boards * ch_per_board
    pedestal = rng.integers(7800, 8201, size=n_ch).astype(np.float32)
    # records dealt round-robin to channels, Poisson gaps (mean 10 us = 1e7 ps) per channel
    ch_index = (np.arange(n_records, dtype=np.int64) % n_ch).astype(np.int32)
    per_ch_rank = np.arange(n_records, dtype=np.int64) // n_ch
    gaps = rng.exponential(1.0e7, size=n_records)
    # cumulative sum per channel: reshape trick on padded (rank, ch) grid
    n_rank = int(per_ch_rank[-1]) + 1 if n_records else 0
    grid = np.zeros((n_rank, n_ch), dtype=np.float64)
    grid.reshape(-1)[: n_records] = gaps
    ts = np.cumsum(grid, axis=0).reshape(-1)[:n_records]
    # each record must not overlap the previous one of its channel: add record span
    ts = ts + per_ch_rank.astype(np.float64) * (L * dt_ns * 1000.0)
    timestamp = ts.astype(np.int64)

    board = (ch_index // ch_per_board).astype(np.int16)
    channel = (ch_index % ch_per_board).astype(np.int16)
    seq = np.arange(n_records, dtype=np.int64)
    pid = np.zeros(n_records, dtype=np.int32)
    order = np.lexsort((seq, channel, board, pid, timestamp))
    return pedestal, ch_index[order], board[order], channel[order], timestamp[order]


def _render_chunk(seed: int, chunk_id: int, ped: np.ndarray, L: int) -> np.ndarray:
    """Render waves for one chunk of records (ped = pedestal per record, float32)."""
    n = len(ped)
    rng = np.random.default_rng([seed, 0x5EED, chunk_id])
    wave = rng.standard_normal((n, L), dtype=np.float32)
    wave *= np.float32(3.0)
    np.rint(wave, out=wave)
    wave += ped[:, None]
    u = rng.random(n)
    n_pulse = np.where(u < 0.15, 2, np.where(u < 0.95, 1, 0))
    t_idx = np.arange(L, dtype=np.float32)[None, :]
    for k in range(2):
        sel = np.flatnonzero(n_pulse > k)
        # draw for every record so the stream does not depend on n_pulse
        start = rng.integers(60, max(L - 120, 61), size=n).astype(np.float32)
        amp = (10.0 ** rng.uniform(1.3, 3.3, size=n)).astype(np.float32)
        tau_f = rng.uniform(10.0, 60.0, size=n).astype(np.float32)
        if sel.size == 0:
            continue
        t = t_idx - start[sel, None]
        np.maximum(t, 0.0, out=t)
        pulse = np.exp(-t / tau_f[sel, None])
        pulse -= np.exp(-t / np.float32(4.0))
        pulse *= amp[sel, None]
        wave[sel] -= pulse
    np.rint(wave, out=wave)
    np.clip(wave, 0.0, 16383.0, out=wave)
    return wave.astype(np.uint16)


def make_run(
    n_records: int,
    preset: str = "v1725",
    *,
    cfg: int = 0,
    L: int | None = None,
    chunk_records: int = 8192,
    threads: int | None = None,
    polarity: str = "unknown",
):
    """Return (records[RECORDS_DTYPE], wave_pool[uint16]) for a synthetic run."""
    pL, dt_ns, n_boards, ch_per_board = PRESETS[preset]
    L = int(L or pL)
    seed = SEED_BASE + int(cfg)
    pedestal, ch_index, board, channel, timestamp = _records_meta(
        n_records, L, dt_ns, n_boards, ch_per_board, seed
    )
    pool = np.empty(n_records * L, dtype=np.uint16)
    pool2d = pool.reshape(n_records, L) if n_records else pool.reshape(0, L)
    ped_rec = pedestal[ch_index]

    n_chunks = (n_records + chunk_records - 1) // chunk_records

    def work(c: int) -> None:
        lo = c * chunk_records
        hi = min(lo + chunk_records, n_records)
        pool2d[lo:hi] = _render_chunk(seed, c, ped_rec[lo:hi], L)

    threads = threads or min(16, os.cpu_count() or 1)
    if n_chunks <= 1 or threads <= 1:
        for c in range(n_chunks):
            work(c)
    else:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(work, range(n_chunks)))

    records = np.zeros(n_records, dtype=RECORDS_DTYPE)
    records["timestamp"] = timestamp
    records["pid"] = 0
    records["board"] = board
    records["channel"] = channel
    records["baseline_upstream"] = np.nan
    records["polarity"] = polarity
    records["record_id"] = np.arange(n_records, dtype=np.int64)
    records["dt"] = dt_ns
    records["wave_offset"] = np.arange(n_records, dtype=np.int64) * L
    records["event_length"] = L
    records["time"] = timestamp // 1000
    if n_records:
        # exact: integer sum / count, as np.mean over float64 of integer samples
        nb = min(BASELINE_SAMPLES, L)
        records["baseline"] = pool2d[:, :nb].sum(axis=1, dtype=np.int64) / float(nb)
    return records, pool


__all__ = ["make_run", "PRESETS", "SEED_BASE", "BASELINE_SAMPLES"]
