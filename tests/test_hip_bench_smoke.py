"""bench.py end to end on a small chunk: the one JSON line the driver parses carries the contract's keys (task statement:
metric / value / unit / n_gpus / steps / warmup / ms_per_step / ..., `roofline`, `cpu_baseline`) and its own parity check
of the slice the CPU baseline ran on."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_small_chunk():
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--records", "40000",
           "--cpu-records", "4000", "--preheat-steps", "10", "--two-sessions"]
    proc = subprocess.run(cmd, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel"].startswith("k_sg_runs32")
    assert r["traffic"] is None                               # the committed PMC capture is of the full-size chunk
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["parity"]["int_fields_bit_exact"] is True and d["parity"]["max_rel_err_float_fields"] < 1e-6
    assert d["clock_ramp"]["preheat_steps"] == 10 and d["clock_ramp"]["first_steps_ms_per_step"] > 0
    assert d["config3"]["total_ms"] > 0 and d["end_to_end"]["rows"] == d["config"]["hits_per_gpu"]
    assert d["two_sessions"]["same_rows"] is True
