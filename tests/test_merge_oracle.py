"""Hit merging: oracle restatement against fixtures produced by the reference's three hit-merge plugins."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd.hit_merge import cluster_bounds, compute_component_rows


@pytest.mark.parametrize("name", G.merge_case_names())
def test_oracle_matches_reference(name):
    case = G.load_merge(name)
    for k, cfg in enumerate(case["configs"]):
        clusters = O.hit_merge_clusters(case["hits"], **cfg)
        G.assert_struct_equal(O.hit_merge_cluster_rows(clusters), case[f"clusters_{k}"], what=f"{name} clusters {k}")
        G.assert_struct_equal(O.hit_merged_rows(case["hits"], clusters), case[f"merged_{k}"], what=f"{name} merged {k}")
        comps = compute_component_rows(case[f"merged_{k}"], case[f"clusters_{k}"])
        G.assert_struct_equal(comps, case[f"components_{k}"], what=f"{name} components {k}")


def test_component_table_checks():
    case = G.load_merge("merge_crafted")
    merged, clusters = case["merged_2"].copy(), case["clusters_2"]
    ids, off = cluster_bounds(clusters)
    assert len(ids) == len(merged) and off[-1] == len(clusters)
    bad = merged.copy()
    bad["component_count"][3] += 1
    with pytest.raises(ValueError, match=r"hit_merged\[3\] component_count mismatch"):
        compute_component_rows(bad, clusters)
    bad = merged.copy()
    bad["component_offset"][1] += 1
    with pytest.raises(ValueError, match=r"hit_merged\[1\] component_offset mismatch"):
        compute_component_rows(bad, clusters)
    with pytest.raises(ValueError, match="cluster count does not match"):
        compute_component_rows(merged[:-1], clusters)
    assert len(compute_component_rows(merged[:0], clusters)) == 0
