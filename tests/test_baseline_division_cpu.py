"""The streaming kernel forms the in-stream baseline as q = tot * (1/40) followed by one FMA correction step
(q + fma(-40, q, tot) * (1/40)); the reference divides (records_builder.py:243-257: mean of the first 40 samples).
Exhaustive proof over every possible sum of 40 uint16 samples that the two agree bit for bit."""

from fractions import Fraction

import numpy as np


def _fma(a: float, b: float, c: float) -> float:
    return float(Fraction(a) * Fraction(b) + Fraction(c))  # float(Fraction) rounds correctly


def test_fma_corrected_reciprocal_equals_division_for_all_sums():
    n = 40
    rn = 1.0 / n
    tot = np.arange(0, n * 65535 + 1, dtype=np.float64)
    exact = tot / float(n)
    q = tot * rn
    # where the plain product is already right the correction must not move it; check a stride of those, and every
    # value where it is wrong (about a third)
    wrong = np.flatnonzero(q != exact)
    assert len(wrong) > 100_000
    right_sample = np.flatnonzero(q == exact)[::37]
    for i in np.concatenate([wrong, right_sample]):
        t, qq = float(tot[i]), float(q[i])
        q1 = _fma(_fma(-float(n), qq, t), rn, qq)
        assert q1 == float(exact[i]), (t, qq, q1, float(exact[i]))
