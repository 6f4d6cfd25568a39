"""find_peaks-based hit detector: oracle vs fixtures produced by the reference's HitFinderPlugin."""

import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G


@pytest.mark.parametrize("name", G.peaks_case_names())
def test_oracle_matches_reference(name):
    case = G.load_peaks(name)
    for k, cfg in enumerate(case["configs"]):
        cfg = dict(cfg)
        pool = case["wave_pool_filtered"] if cfg.pop("use_filtered", True) else case["wave_pool"]
        got = O.find_peak_hits(case["records"], pool, **cfg)
        G.assert_struct_equal(got, case[f"hit_{k}"], what=f"{name} cfg {k}")
