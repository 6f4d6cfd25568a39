"""HipHitMergeClustersPlugin / HipHitMergePlugin / HipHitMergedComponentsPlugin -- drop-ins for the three
hit-merge plugins (reference: waveform_analysis/core/plugins/builtin/cpu/hit_merge.py:325-544)."""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import hit_merge as M
from ..dtypes import HIT_MERGE_CLUSTERS_DTYPE, HIT_MERGED_COMPONENTS_DTYPE, HIT_MERGED_DTYPE
from ..plugin_api import Option, Plugin
from . import _common as K

MERGE_OPTIONS = {
    "merge_gap_ns": Option(default=0.0, type=float, help="largest edge-to-edge gap (ns) that still merges; <= 0: no merging"),
    "max_total_width_ns": Option(default=10000.0, type=float, help="largest total width (ns) of a chained cluster"),
    "dt": Option(default=None, type=int, help="sample interval (ns) when hit_threshold lacks dt"),
}


def _session(context):
    pool_obj = getattr(context, "wfa_device_pool", None) or K.default_pool()
    return K.note_session(pool_obj.session())


def _clusters_or_compute(context, run_id, hits, cfg_plugin):
    """hit_merged and hit_merged_components take hit_merge_clusters from the context when it can provide it and
    recompute it otherwise (hit_merge.py:363-380, 449-480)."""
    try:
        rows = context.get_data(run_id, "hit_merge_clusters")
    except Exception:
        rows = None
    if rows is None:
        gap, width, explicit_dt = M.resolve_merge_config(context, cfg_plugin)
        rows = M.compute_cluster_rows(_session(context), hits, gap, width, explicit_dt, "hit_merge_clusters")
    return rows


class HipHitMergeClustersPlugin(K.HipPlugin):
    """Flat cluster membership of the per-channel hit merge."""

    provides = "hit_merge_clusters"
    algorithmic_bytes = (0, 0, 60)  # hit rows in, index tables out
    depends_on = ["hit_threshold"]
    description = "Internal cluster membership rows shared by hit_merged outputs (HIP, gfx950)."
    version = "0.1.0+hip1"
    save_when = "always"
    output_dtype = HIT_MERGE_CLUSTERS_DTYPE
    options = MERGE_OPTIONS

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        hits = context.get_data(run_id, "hit_threshold")
        if not isinstance(hits, np.ndarray):
            raise ValueError("hit_merge_clusters expects hit_threshold as a single structured array")
        if len(hits) == 0:
            return np.zeros(0, dtype=HIT_MERGE_CLUSTERS_DTYPE)
        gap, width, explicit_dt = M.resolve_merge_config(context, self)
        return M.compute_cluster_rows(_session(context), hits, gap, width, explicit_dt, self.provides)


class HipHitMergePlugin(K.HipPlugin):
    """Merge nearby hits from hit_threshold within the same channel."""

    provides = "hit_merged"
    algorithmic_bytes = (0, 0, 60)  # hit rows in, index tables out
    depends_on = ["hit_threshold", "hit_merge_clusters"]
    description = "Merge nearby threshold hits per channel with time-gap and max-width constraints (HIP, gfx950)."
    version = "0.8.0+hip1"
    save_when = "always"
    output_dtype = HIT_MERGED_DTYPE
    options = MERGE_OPTIONS

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        hits = context.get_data(run_id, "hit_threshold")
        if not isinstance(hits, np.ndarray):
            raise ValueError("hit_merged expects hit_threshold as a single structured array")
        if len(hits) == 0:
            return np.zeros(0, dtype=HIT_MERGED_DTYPE)
        _gap, _width, explicit_dt = M.resolve_merge_config(context, self)
        cluster_rows = _clusters_or_compute(context, run_id, hits, self)
        if not isinstance(cluster_rows, np.ndarray):
            raise ValueError("hit_merged expects hit_merge_clusters as a structured array")
        return M.compute_merged_rows(_session(context), hits, cluster_rows, explicit_dt, self.provides)


class HipHitMergedComponentsPlugin(Plugin):
    """Flat component hit indices of every hit_merged row."""

    provides = "hit_merged_components"
    depends_on = ["hit_merge_clusters", "hit_merged"]
    description = "Return per-cluster component hit indices for hit_merged rows."
    version = "0.1.0+hip1"
    save_when = "always"
    output_dtype = HIT_MERGED_COMPONENTS_DTYPE

    def compute(self, context: Any, run_id: str, **_kwargs) -> np.ndarray:
        merged = context.get_data(run_id, "hit_merged")
        if not isinstance(merged, np.ndarray):
            raise ValueError("hit_merged_components expects hit_merge_clusters and hit_merged structured arrays")
        if len(merged) == 0:
            return np.zeros(0, dtype=HIT_MERGED_COMPONENTS_DTYPE)
        try:
            cluster_rows = context.get_data(run_id, "hit_merge_clusters")
        except Exception:
            cluster_rows = None
        if cluster_rows is None:
            hits = context.get_data(run_id, "hit_threshold")
            cluster_rows = _clusters_or_compute(context, run_id, hits, context.get_plugin("hit_merged"))
        if not isinstance(cluster_rows, np.ndarray):
            raise ValueError("hit_merged_components expects hit_merge_clusters and hit_merged structured arrays")
        return M.compute_component_rows(merged, cluster_rows)
