"""HipHitGroupedPlugin -- event grouping endpoint of the path
(reference: HitGroupedPlugin, waveform_analysis/core/plugins/builtin/cpu/event_analysis.py:69-106)."""

from __future__ import annotations

from typing import Any

from ..event_grouping import group_hit_windows
from ..plugin_api import Option, Plugin
from . import _common as K


class HipHitGroupedPlugin(K.HipPlugin):
    """Group hits across channels into coincidence events (gap-chained absolute windows).

    Consumes `hit_merged` (+ `hit_merged_components` and `hit_threshold` for merged hits that span
    records, as the reference does) when that product is registered and otherwise the threshold hits
    directly: with the reference's default `merge_gap_ns = 0` nothing is merged and both give the
    same events.  On several GPUs the rows of all ranks are gathered first (sharding.py).
    """

    provides = "hit_grouped"
    algorithmic_bytes = (0, 0, 72)  # device pass: bytes per sample, per record, per output row
    depends_on = []  # dynamic, see resolve_depends_on
    description = "Group hits across channels into event-level coincidence windows (vectorised)."
    version = "0.5.0+hip1"
    save_when = "always"
    options = {
        "time_window_ns": Option(default=100.0, type=float),
        "dt": Option(default=None, type=int, help="sample interval (ns) when the hit rows lack dt"),
    }

    def resolve_depends_on(self, context: Any, run_id: str | None = None) -> list[str]:
        plugins = getattr(context, "_plugins", {}) or {}
        data = getattr(context, "_data", {}) or {}
        if "hit_merged" in plugins or "hit_merged" in data:
            return ["hit_merged", "hit_merged_components", "hit_threshold"]
        return ["hit_threshold"]

    def compute(self, context: Any, run_id: str, **kwargs) -> Any:
        source = self.resolve_depends_on(context, run_id)[0]
        hits = context.get_data(run_id, source)
        time_window_ns = float(context.get_config(self, "time_window_ns"))
        explicit_dt = K.resolve_dt_config(context, self, deprecated_keys=("sampling_interval_ns", "dt_ns"))
        dt_values = K.require_dt_array(hits, explicit_dt=explicit_dt, plugin_name=self.provides, data_name=source)
        component_rows = component_hits = None
        if source == "hit_merged":
            component_rows = context.get_data(run_id, "hit_merged_components")
            component_hits = context.get_data(run_id, "hit_threshold")
        pool_obj = getattr(context, "wfa_device_pool", None) or K.default_pool()
        return group_hit_windows(hits, time_window_ns=time_window_ns, dt_values=dt_values,
                                 component_rows=component_rows, component_hits=component_hits,
                                 session=K.note_session(pool_obj.session()))
