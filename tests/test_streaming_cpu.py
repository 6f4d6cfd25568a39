"""Streaming driver (chunking, halo, clipping, ordering) against what the reference's StreamingPlugin does with the same
records table; the identity plugin needs no GPU."""

import json
import os

import numpy as np
import pytest

from tests import golden_util as G
from waveformanalysis_amd.plugin_api import SimpleContext
from waveformanalysis_amd.streaming import HipStreamingPlugin


class Identity(HipStreamingPlugin):
    provides = "ident"
    depends_on = ["records"]
    chunk_size = 64
    length_field = "event_length"
    parallel = False


def load():
    z = np.load(os.path.join(G.GOLDEN, "chunk_streaming.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["options"] = json.loads(bytes(d.pop("options_json")).decode())
    return d


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
@pytest.mark.parametrize("parallel", [False, True])
def test_chunks_match_reference(tag, parallel):
    case = load()
    rec = case["records"]
    kw = case["options"][tag]
    p = Identity()
    p._apply_streaming_config(kw)
    chunks = list(p._data_to_chunks(rec, "run"))
    got = [(c.start, c.end, c.metadata["main_start"], c.metadata["main_end"], c.metadata["segment_id"], len(c),
            int(c.data["record_id"][0]), int(c.data["record_id"][-1])) for c in chunks]
    np.testing.assert_array_equal(got, case[f"{tag}_in"])
    ctx = SimpleContext({}, {"records": rec})
    res = list(p.compute(ctx, "run", streaming_config={**kw, "parallel": parallel, "max_workers": 3}))
    got = [(c.start, c.end, len(c), int(c.data["record_id"][0]), int(c.data["record_id"][-1])) for c in res]
    np.testing.assert_array_equal(got, case[f"{tag}_out"])
    assert sum(len(c) for c in res) == len(rec) or kw.get("clip_strict")


def test_driver_rules():
    case = load()
    rec = case["records"]
    ctx = SimpleContext({}, {"records": rec})

    class Stateful(Identity):
        is_stateful = True
        parallel = True

        def __init__(self):
            self.resets = 0

        def reset_state(self):
            self.resets += 1

    s = Stateful()
    out = list(s.compute(ctx, "run"))
    assert s.resets == 3 and len(out) == len(case["a_out"])      # one reset per time segment, serial order

    class Dropper(Identity):
        def compute_chunk(self, chunk, context, run_id, **kw):
            return None if chunk.metadata["segment_id"] == 1 else chunk.data    # plain arrays are wrapped

    res = list(Dropper().compute(ctx, "run"))
    assert all(c.metadata["segment_id"] != 1 for c in res) and all(c.data_type == "ident" for c in res)

    class Leaky(Identity):
        def compute_chunk(self, chunk, context, run_id, **kw):
            bad = chunk.data.copy()
            bad["event_length"][-1] = 10**9     # the row now ends far beyond the chunk
            return bad

    with pytest.raises(ValueError, match="ends at"):
        list(Leaky().compute(ctx, "run"))
    with pytest.warns(UserWarning, match="Unknown streaming_config keys"):
        list(Identity().compute(ctx, "run", streaming_config={"nope": 1}))
    with pytest.raises(TypeError, match="streaming_config must be a dict"):
        list(Identity().compute(ctx, "run", streaming_config=[1]))
    assert list(Identity().compute(SimpleContext({}, {"records": rec[:0]}), "run")) == []
