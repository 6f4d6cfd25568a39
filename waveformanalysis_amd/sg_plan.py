"""Host-side Savitzky-Golay plan for the HIP kernels.

The reference filters every record with
``scipy.signal.savgol_filter(x_f32, W, P, mode="interp")`` (waveform_analysis/core/plugins/
builtin/cpu/filtering.py:226-240), where W is clamped to the record length and made odd
(filtering.py:181-195).  The kernels need, for every effective window w = 1, 3, ..., W:

* ``fw``      the float64 correlation weights scipy convolves with (``savgol_coeffs`` reversed),
              plus whether ``ndimage.correlate1d`` takes its symmetric branch for them;
* ``E_left / E_right``  the rows of the least-squares projection ("hat") matrix that scipy's
              ``_fit_edges_polyfit`` evaluates for the first / last w//2 samples;
* for the full window, an *integer* plan: the hat matrix is rational, H = N / den with integer
  N, so for uint16 samples ``y = (N . x) / den`` is an exact rational.  The kernels accumulate
  N . x in int32 on the full-rate integer VALU and round once; DESIGN.md section "exact integer
  Savitzky-Golay" proves that the float32 result equals scipy's whenever |N . x| >= guard, and
  the kernels evaluate scipy's float64 chain literally below the guard.

Everything here is exact rational arithmetic (fractions.Fraction); scipy is consulted only for
its own coefficient bits (they come out of an lstsq and differ from the correctly rounded
rationals by a few ulp, which matters only below the guard).
"""

from __future__ import annotations

from dataclasses import dataclass
from fractions import Fraction
from functools import lru_cache
import math

import numpy as np

MAX_WINDOW = 63  # WFA_MAX_SG_WINDOW
X_MAX = 65535    # uint16 samples
DBL_EPSILON = 2.220446049250313e-16


def normalize_window(sg_window_size: int, sg_poly_order: int) -> tuple[int, int]:
    """Plugin-level validation, filtering.py:105-118 (even window -> +1)."""
    window = int(sg_window_size)
    order = int(sg_poly_order)
    if window <= 0:
        raise ValueError(f"SG 窗口大小 ({window}) 必须大于 0")
    if order < 0:
        raise ValueError(f"SG 多项式阶数 ({order}) 必须大于等于 0")
    if window % 2 == 0:
        window += 1
    if order >= window:
        raise ValueError(f"SG 多项式阶数 ({order}) 必须小于窗口大小 ({window})")
    return window, order


def hat_matrix(w: int, p: int) -> list[list[Fraction]]:
    """Exact projection onto polynomials of degree <= p sampled at 0..w-1 (w x w)."""
    n = p + 1
    V = [[Fraction(i) ** k for k in range(n)] for i in range(w)]
    A = [[sum(V[i][a] * V[i][b] for i in range(w)) for b in range(n)] for a in range(n)]
    M = [row[:] + [Fraction(int(i == j)) for j in range(n)] for i, row in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if M[r][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        inv = 1 / M[c][c]
        M[c] = [v * inv for v in M[c]]
        for r in range(n):
            if r != c and M[r][c] != 0:
                f = M[r][c]
                M[r] = [a - f * b for a, b in zip(M[r], M[c])]
    Ainv = [row[n:] for row in M]
    VA = [[sum(V[i][a] * Ainv[a][b] for a in range(n)) for b in range(n)] for i in range(w)]
    return [[sum(VA[i][b] * V[j][b] for b in range(n)) for j in range(w)] for i in range(w)]


def _lcm_den(rows) -> int:
    d = 1
    for row in rows:
        for v in row:
            d = d * v.denominator // math.gcd(d, v.denominator)
    return d


def _scipy_coeffs(w: int, p: int):
    try:
        from scipy.signal import savgol_coeffs
    except Exception:  # scipy absent: correctly rounded rationals
        return None
    return np.asarray(savgol_coeffs(w, p), dtype=np.float64)


def _is_symmetric(fw: np.ndarray) -> bool:
    """ndimage NI_Correlate1D symmetry test (|fw[c+i] - fw[c-i]| <= DBL_EPSILON for all i)."""
    w = len(fw)
    if w % 2 == 0:
        return False
    c = w // 2
    return all(abs(fw[c + i] - fw[c - i]) <= DBL_EPSILON for i in range(1, c + 1))


@dataclass(frozen=True)
class SgPlan:
    window: int
    polyorder: int
    tab: np.ndarray        # float64 [n_tables * stride]
    symmetric: np.ndarray  # uint8 [n_tables]
    int_ok: bool
    itab: np.ndarray       # int32 [stride]
    den: int
    den_edge: int
    guard: int
    guard_edge: int

    @property
    def n_tables(self) -> int:
        return (self.window + 1) // 2

    @property
    def stride(self) -> int:
        return self.window + 2 * (self.window // 2) * self.window


@lru_cache(maxsize=32)
def build_plan(sg_window_size: int = 11, sg_poly_order: int = 2) -> SgPlan:
    W, P = normalize_window(sg_window_size, sg_poly_order)
    if W > MAX_WINDOW:
        raise ValueError(f"SG window {W} exceeds the supported maximum {MAX_WINDOW}")
    H = W // 2
    n_tables = (W + 1) // 2
    stride = W + 2 * H * W
    tab = np.zeros(n_tables * stride, dtype=np.float64)
    sym = np.ones(n_tables, dtype=np.uint8)

    full_hat = None
    fw_full = None
    for t in range(n_tables):
        w = 2 * t + 1
        if w <= P:
            continue  # filter is a copy for such short records
        hat = hat_matrix(w, P)
        h = w // 2
        coeffs = _scipy_coeffs(w, P)
        if coeffs is None:
            coeffs = np.array([float(v) for v in hat[h]], dtype=np.float64)[::-1]
        fw = coeffs[::-1].copy()  # convolve1d correlates with the reversed kernel
        base = t * stride
        tab[base : base + w] = fw
        sym[t] = 1 if _is_symmetric(fw) else 0
        for i in range(h):
            tab[base + W + i * W : base + W + i * W + w] = [float(v) for v in hat[i]]
            tab[base + W + H * W + i * W : base + W + H * W + i * W + w] = [float(v) for v in hat[w - h + i]]
        if w == W:
            full_hat, fw_full = hat, fw

    # ---- integer plan for the full window ----------------------------------------------------
    itab = np.zeros(stride, dtype=np.int32)
    int_ok = False
    den = den_edge = 1
    guard = guard_edge = 0
    if full_hat is not None and W > P:
        center = full_hat[H]
        den = _lcm_den([center])
        n_center = [int(v * den) for v in center]
        edge_rows = full_hat[:H] + full_hat[W - H :]
        den_edge = _lcm_den(edge_rows) if edge_rows else 1
        n_edge = [[int(v * den_edge) for v in row] for row in edge_rows]
        lim = 2**31 - 1
        fits = sum(abs(v) for v in n_center) * X_MAX < lim and all(
            sum(abs(v) for v in row) * X_MAX < lim for row in n_edge
        )
        if fits and den < 2**24 and den_edge < 2**24:
            int_ok = True
            itab[:W] = n_center[::-1]  # same orientation as fw (symmetric anyway)
            for i in range(H):
                itab[W + i * W : W + i * W + W] = n_edge[i]
                itab[W + H * W + i * W : W + H * W + i * W + W] = n_edge[H + i]
            # scipy's float64 chain deviates from the exact rational by at most eps_c * X_MAX;
            # float32 rounding of both agrees while |y| >= 2 * eps * den * 2**24.
            exact = np.array([float(v) for v in center], dtype=np.float64)[::-1]
            sum_abs = float(sum(abs(v) for v in center))
            eps_c = ((W + 2) * 2.0**-53 * sum_abs + float(np.sum(np.abs(fw_full - exact)))) * X_MAX
            guard = int(math.ceil(4 * 2 * eps_c * den * den * 2.0**24)) + 1
            sum_abs_e = max(float(sum(abs(v) for v in row)) for row in edge_rows) if edge_rows else 0.0
            eps_e = 1024 * 2.0**-53 * sum_abs_e * X_MAX
            guard_edge = int(math.ceil(4 * 2 * eps_e * den_edge * den_edge * 2.0**24)) + 1
    return SgPlan(W, P, tab, sym, int_ok, itab, den, den_edge, guard, guard_edge)


__all__ = ["SgPlan", "build_plan", "normalize_window", "hat_matrix", "MAX_WINDOW"]
