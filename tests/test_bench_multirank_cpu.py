"""bench.py's N > 1 path rehearsed on CPU: `python bench.py --gpus 2` with no launcher around it starts its two ranks
itself (fresh child processes), they rendezvous over gloo on 127.0.0.1, time, gather, and rank 0 prints the one JSON line;
a failed exchange still prints the line, says gather_ok = false and makes every rank exit non-zero (VERDICT r1 item 5,
ADVICE r1: a dead transport used to yield rc 0).  The device is a stand-in (tests/bench_stub.py): what is under test is
bench.py's own control flow."""

import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, port):
    env = dict(os.environ, WFA_BENCH_STUB="tests.bench_stub", PYTHONPATH=REPO, WFA_BENCH_GATHER_TIMEOUT_S="60")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--records", "2000", "--steps", "2", "--warmup", "1",
           "--no-features", "--no-cpu-baseline", "--master-port", str(port)]
    return subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_gpus_2_self_launch_prints_one_line_and_exits_zero():
    p = _run({}, 29713)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["gather_ok"] is True
    assert d["config"]["samples_per_gpu"] == 2000 * 800
    assert abs(d["value"] - 2 * 1.6e6 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-3 * d["value"] + 1e-9   # whole-job aggregate
    assert d["gather"]["hits_total"] == 1000 + 1001 and d["gather"]["events"] is not None


def test_failed_gather_is_reported_and_exits_nonzero():
    p = _run({"WFA_BENCH_STUB_FAIL_GATHER": "1"}, 29714)
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (p.stdout, p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["gather_ok"] is False and d["n_gpus"] == 2
