"""V1725 DAW_DEMO binary -> records + wave_pool: header walk (host C), order + payload packing (GPU) against a fixture
produced by the reference's V1725Reader / build_records_from_v1725_files."""

import os

import numpy as np
import pytest

from oracle import wfa_oracle as O
from tests import golden_util as G


def load():
    z = np.load(os.path.join(G.GOLDEN, "v1725bin_files.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def test_header_walk_matches_reference_reader():
    """No GPU needed: wfa_v1725_index is host code of the library."""
    from waveformanalysis_amd.records_builder import v1725_index

    case = load()
    for k in range(3):
        idx = v1725_index(case[f"blob{k}"])
        want = case[f"index{k}"]
        got = np.stack([idx["channel"], idx["timestamp"], idx["trunc"], idx["baseline"], idx["n_samples"]], axis=1)
        np.testing.assert_array_equal(got, want)
        waves = O.v1725_waves(bytes(case[f"blob{k}"]))
        assert len(waves) == len(want)
        blob = case[f"blob{k}"]
        for w, off, n in zip(waves, idx["payload_offset"], idx["n_samples"]):
            np.testing.assert_array_equal(blob[off : off + 2 * n].view(np.int16), w[4])
    assert len(v1725_index(np.zeros(0, dtype=np.uint8))["channel"]) == 0
    assert len(v1725_index(case["blob0"][:15])["channel"]) == 0          # short event header
    bad = case["blob0"].copy()
    bad[16:19] = (2, 0, 0)                                                 # channel size below the 3 header words
    with pytest.raises(ValueError, match="channel size"):
        v1725_index(bad)


def test_oracle_matches_reference_build():
    case = load()
    names = bytes(case["names"]).decode().split("\n")
    boards = [int(n.split("_b")[1].split("_")[0]) for n in names]
    rec, pool = O.build_records_from_v1725_blobs([case[f"blob{k}"] for k in range(3)], boards, 4)
    G.assert_struct_equal(rec, case["records"])
    np.testing.assert_array_equal(pool, case["wave_pool"])
    rec, pool = O.build_records_from_v1725_blobs([case["blob2"]], boards[2:], 2)
    G.assert_struct_equal(rec, case["records_single"])
    np.testing.assert_array_equal(pool, case["wave_pool_single"])


@pytest.mark.gpu
def test_gpu_build_from_files(tmp_path):
    from waveformanalysis_amd.records_builder import build_records_from_v1725_files

    case = load()
    names = bytes(case["names"]).decode().split("\n")
    paths = []
    for k, name in enumerate(names):
        p = tmp_path / name
        p.write_bytes(bytes(case[f"blob{k}"]))
        paths.append(str(p))
    b = build_records_from_v1725_files(paths, dt_ns=4)
    G.assert_struct_equal(b.records, case["records"])
    np.testing.assert_array_equal(b.wave_pool, case["wave_pool"])
    b = build_records_from_v1725_files(paths[2:] + [str(tmp_path / "missing_b3_seg0.bin")], dt_ns=2)
    G.assert_struct_equal(b.records, case["records_single"])
    np.testing.assert_array_equal(b.wave_pool, case["wave_pool_single"])
    assert len(build_records_from_v1725_files([], dt_ns=4).records) == 0
