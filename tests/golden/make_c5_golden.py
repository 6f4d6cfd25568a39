"""Generate tests/golden/c5_replay.npz: BASELINE.json's config 5 run by the REFERENCE (build container only).

Usage (from any scratch cwd; the reference tree is never written to):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_c5_golden.py

The two config-5 chains (waveformanalysis_amd/replay.py: c5_chain) are declared as strax-shaped classes around the
reference's own CPU plugin classes and driven by the reference's own StraxPluginAdapter / StraxContextAdapter
(core/plugins/core/adapters.py:28-440) in front of a real reference Context:
    FilteredWaveformsPlugin   cpu/filtering.py:410-536          BasicFeaturesPlugin   cpu/basic_features.py:43-278
    HitFinderPlugin           cpu/peak_finding.py:446-614       WaveformWidthPlugin   cpu/waveform_width.py:97-374
    S1S2ClassifierPlugin      cpu/s1_s2_classifier.py:133-228   ThresholdHitPlugin    cpu/hit_finder.py:82-413
    HitMergeClustersPlugin / HitMergePlugin  cpu/hit_merge.py:325-445   HitGroupedPlugin  cpu/event_analysis.py:69-
The input is the seeded synthetic VX2730-like run of SURVEY 8d (256 channels, L = 1500, 10^7 samples), mirrored to
positive-going pulses (replay.mirror_positive: the reference's width stage drops peaks below the baseline): only the
generator arguments are stored, plus the reference's output tables.  The fixture is data only.
"""

from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)
os.chdir(tempfile.mkdtemp(prefix="wfa_c5_golden_"))
sys.dont_write_bytecode = True

from waveform_analysis.core.context import Context  # noqa: E402
from waveform_analysis.core.plugins.builtin.cpu.basic_features import BasicFeaturesPlugin  # noqa: E402
from waveform_analysis.core.plugins.builtin.cpu.event_analysis import HitGroupedPlugin  # noqa: E402
from waveform_analysis.core.plugins.builtin.cpu.filtering import FilteredWaveformsPlugin  # noqa: E402
from waveform_analysis.core.plugins.builtin.cpu.hit_finder import ThresholdHitPlugin  # noqa: E402
from waveform_analysis.core.plugins.builtin.cpu.hit_merge import (  # noqa: E402
    HitMergeClustersPlugin,
    HitMergedComponentsPlugin,
    HitMergePlugin,
)
from waveform_analysis.core.plugins.builtin.cpu.peak_finding import HitFinderPlugin  # noqa: E402
from waveform_analysis.core.plugins.builtin.cpu.s1_s2_classifier import S1S2ClassifierPlugin  # noqa: E402
from waveform_analysis.core.plugins.builtin.cpu.waveform_width import WaveformWidthPlugin  # noqa: E402
from waveform_analysis.core.plugins.core.adapters import StraxContextAdapter, StraxPluginAdapter  # noqa: E402

from waveformanalysis_amd import replay, synth  # noqa: E402

N_RECORDS, PRESET, CFG = 6672, "vx2730", 5      # 6672 x 1500 = 1.0008e7 samples

assert replay.StraxPluginAdapter is StraxPluginAdapter and replay.StraxContextAdapter is StraxContextAdapter

REFERENCE_PLUGINS = {
    "filtered_waveforms": FilteredWaveformsPlugin, "basic_features": BasicFeaturesPlugin, "hit": HitFinderPlugin,
    "waveform_width": WaveformWidthPlugin, "s1_s2": S1S2ClassifierPlugin, "hit_threshold": ThresholdHitPlugin,
    "hit_merge_clusters": HitMergeClustersPlugin, "hit_merged": HitMergePlugin,
    "hit_merged_components": HitMergedComponentsPlugin, "hit_grouped": HitGroupedPlugin,
}


def main():
    rec, pool = replay.mirror_positive(*synth.make_run(N_RECORDS, PRESET, cfg=CFG))
    ctx = Context(storage_dir=os.path.join(os.getcwd(), "store"))
    tables = replay.replay_c5(rec, pool, plugins=REFERENCE_PLUGINS, context=ctx)
    out = {name: np.asarray(tables[name]) for name in replay.C5_TARGETS if name != "hit_grouped"}
    out.update(replay.flatten_grouped(tables["hit_grouped"]))
    assert all(v.dtype != object for v in out.values())
    out["options_json"] = np.frombuffer(json.dumps({"n_records": N_RECORDS, "preset": PRESET, "cfg": CFG}).encode(), dtype=np.uint8)
    path = os.path.join(REPO, "tests", "golden", "c5_replay.npz")
    np.savez_compressed(path, **out)
    labels = np.bincount(out["s1_s2"]["label"], minlength=3).tolist() if "label" in out["s1_s2"].dtype.names else None
    print({k: len(v) for k, v in out.items() if k != "options_json" and not k.startswith("grouped_") or k == "grouped_t_min"},
          "labels", labels, "->", os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
