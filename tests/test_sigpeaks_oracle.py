"""SignalPeaksStreamPlugin: the host chunk rule of the HIP plugin and the oracle's per-row restatement against the
reference plugin's own chunks and rows (tests/golden/sigpeaks_*.npz).  No device needed."""

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.plugins import HipSignalPeaksStreamPlugin


def _inputs(case, cfg):
    cfg = dict(cfg)
    sc = cfg.pop("streaming_config")
    plugin = HipSignalPeaksStreamPlugin()
    plugin._apply_streaming_config({"parallel": False, **sc})
    ctx = SimpleContext(cfg, {"st_waveforms": case["st_waveforms"], "filtered_waveforms": case["filtered_waveforms"]})
    plugin._load_config(ctx)
    return plugin, ctx, cfg, list(plugin._get_input_chunks(ctx, "run"))


@pytest.mark.parametrize("name", G.sigpeaks_case_names())
def test_input_chunks_match_reference(name):
    case = G.load_sigpeaks(name)
    for k, cfg in enumerate(case["configs"]):
        _p, _c, _cfg, ins = _inputs(case, cfg)
        got = np.array([(c.start, c.end, len(c.data), c.metadata["event_offset"], c.metadata["channel_index"],
                         c.metadata["segment_id"], int(c.dt)) for c in ins], dtype=np.int64)
        np.testing.assert_array_equal(got, case[f"inputs_{k}"], err_msg=f"{name} cfg {k}")


@pytest.mark.parametrize("name", G.sigpeaks_case_names())
def test_oracle_rows_match_reference(name):
    case = G.load_sigpeaks(name)
    for k, cfg in enumerate(case["configs"]):
        plugin, _ctx, opts, ins = _inputs(case, cfg)
        outs = []
        for c in ins:
            rows = O.signal_peaks_rows(c.data, c.metadata["filtered_waveforms"], event_offset=c.metadata["event_offset"],
                                       **{a: b for a, b in opts.items()})
            res = plugin._postprocess_result(
                None if len(rows) == 0 else type(c)(rows, int(rows["timestamp"].min()), int(rows["timestamp"].max()),
                                                    run_id="run", data_type="signal_peaks_stream", data_kind="peaks",
                                                    time_field="timestamp"), c)
            if res is not None:
                outs.append(res)
        got = np.array([(c.start, c.end, len(c.data)) for c in outs], dtype=np.int64).reshape(-1, 3)
        np.testing.assert_array_equal(got, case[f"chunks_{k}"], err_msg=f"{name} cfg {k} chunks")
        G.assert_struct_equal(np.concatenate([c.data for c in outs]), case[f"rows_{k}"], what=f"{name} cfg {k}")
