"""AddressSanitizer + UndefinedBehaviorSanitizer run of the host-only C++ of libwfa_hip.so (csrc/wfa_host.hpp: the V1725
header walk and the pinned staging ring of the pool uploads) behind a stand-in device (csrc/host_check.cpp) -- SURVEY
section 5 assigns this build to the backend (the reference has none); CPU box only, the GPU pool has no sanitizer.

The V1725 streams are the reference-made fixture (tests/golden/v1725bin_files.npz: blobs written by the generator, index
tables read back by the reference's V1725Reader); the digests the driver prints are recomputed here from those tables."""

import json
import os
import subprocess

import numpy as np
import pytest

from tests import golden_util as G

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "waveformanalysis_amd", "csrc")
BIN = os.path.join(CSRC, "build_tmp", "host_check_asan")


@pytest.fixture(scope="module")
def host_check():
    res = subprocess.run(["make", "-C", CSRC, "SANITIZE=1", "host_check"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert os.path.exists(BIN)
    return BIN


def run(binary, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run([binary, *map(str, args)], capture_output=True, text=True, timeout=600, env=env)
    assert "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-4000:]
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_v1725_header_walk_under_sanitizers(host_check, tmp_path):
    case = np.load(os.path.join(G.GOLDEN, "v1725bin_files.npz"), allow_pickle=False)
    for k in range(3):
        blob, want = case[f"blob{k}"], case[f"index{k}"]     # columns: channel, timestamp, trunc, baseline, n_samples
        path = tmp_path / f"blob{k}.bin"
        path.write_bytes(blob.tobytes())
        got = run(host_check, "v1725", path)
        n = len(want)
        assert got["waves"] == n and got["samples"] == int(want[:, 4].sum())
        assert got["channel_sum"] == int(want[:, 0].sum()) and got["baseline_sum"] == int(want[:, 3].sum())
        assert got["ts_digest"] == int((want[:, 1].astype(np.uint64) * np.arange(1, n + 1, dtype=np.uint64)).sum())
    # the product's entry point runs the same function (wfa_hits.hip includes wfa_host.hpp)
    assert "host::v1725_index" in open(os.path.join(CSRC, "wfa_hits.hip")).read()
    assert "host::staged_copy" in open(os.path.join(CSRC, "wfa_capi.hip")).read()


@pytest.mark.parametrize("nbytes,stage", [(50_000_000, 8 << 20), ((8 << 20) * 3, 8 << 20), (5, 4096), (1 << 20, 1 << 20),
                                          ((1 << 20) + 1, 1 << 20), (0, 4096)])
def test_staging_ring_under_sanitizers(host_check, nbytes, stage):
    """Two staging buffers, asynchronous copies out of them: whole chunks, one byte over, less than a chunk, nothing."""
    got = run(host_check, "ring", nbytes, stage)
    assert got["bytes"] == nbytes and got["chunks"] == (nbytes + stage - 1) // stage
