"""bench.py's config-4 leg (the N > 1 part of the bench: 256-channel VX2730 run, fused pass on the padded streaming route,
rows gathered over RCCL, the gathered table verified against the ranks' digests, grouped from the device buffer) on the one
GPU there is: a world of one rank -- real gloo control plane, real 1-rank RCCL communicator, the function the driver's
multi-GPU run calls."""

import argparse
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_config4_leg_world_of_one():
    from waveformanalysis_amd import _lib

    _lib.load()  # /opt/rocm's HIP + RCCL first, as bench.py does (torch bundles its own copies)
    import torch.distributed as dist

    import bench
    from waveformanalysis_amd.device import DeviceSession

    dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        args = argparse.Namespace(c4_records=24_000, threshold=10.0)
        out = bench.config4_leg(args, DeviceSession, dist, 0, 1, 0)
    finally:
        dist.destroy_process_group()
    assert out["ok"] is True and out["verified"] is True, out
    assert out["samples_per_gpu"] == 24_000 * 1500 and out["hits_total"] > 24_000
    assert out["channels_seen"] == 256 and 0 < out["events"] <= out["hits_total"]
    assert out["gather_ms"] > 0 and out["group_hit_windows_ms"] > 0
