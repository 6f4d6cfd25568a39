#!/usr/bin/env python3
"""bench.py -- Gsamples/s of the fused baseline + Savitzky-Golay filter + threshold-hit pass.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--records R] [--preset v1725|vx2730]

One *step* = one pass of the hot path over one resident chunk: the streaming kernel (baseline estimate over the first 40
samples, SG(11,2) filter evaluated on the fly, threshold mask -> ordered run events), the scan of the per-span hit
counts, the descriptor gather and the hit-row kernels.  Inputs (wave_pool uint16, records SoA) are resident in HBM
before the timed region; hit rows stay on the device.

N > 1: one rank per GPU.  Launched either by the driver (`python -m torch.distributed.run ... bench.py --gpus N`, ranks
read RANK / LOCAL_RANK / WORLD_SIZE) or plainly as `python bench.py --gpus N`: then this process starts the N ranks
itself as a child `torch.distributed.run` -- before anything in it has touched HIP -- relays their JSON line and exits
with their status.  Records shard by channel with no data-path collective: every rank processes
its own equally sized shard ("weak" scaling), value = all samples / max-over-ranks time.  After the timed region the
ranks' hit rows are gathered to rank 0 over RCCL (the event-grouping exchange), stay on its device, and are grouped
there (`group_hit_windows`, 100 ns) without a host round trip; `gather_ok` says whether that exchange completed -- if
it did not, the line is still printed and the process exits non-zero.

Clock state: W warm-up steps (the first GPU work after setup, timed as `clock_ramp.first_steps_ms_per_step`), then
`--preheat-steps` untimed queued passes (~100 ms: what the GPU needs to reach its sustained clock from idle -- a handful of
sub-millisecond warm-up steps does not get it there), then the K timed steps.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit..., plus
  roofline     achieved = (2*N + 29*R + 60*H) bytes / mean duration of the streaming kernel, measured with HIP events
               on the kernel's own stream inside the timed region; peak 8000 GB/s; `frac_pass` = the same bytes over
               the whole step; `traffic` = HBM bytes per launch from rocprofv3 PMC passes of THIS kernel on THIS
               workload (profiles/hbm_traffic.json, keyed by kernel | preset | records | L), else null
  cpu_baseline the oracle's port of the reference loops on a bounded slice of the same chunk on this host: the filter
               leg on os.cpu_count() threads (the reference's default, cpu/records.py:405-430), hit finding on one
  end_to_end   HipThresholdHitPlugin.compute on the same chunk from host arrays: H2D + kernels + D2H of the rows
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# dominant (streaming) kernel of the fused pass, by preference: uniform records on the run-event kernel, else the
# bitmap route's mask kernels
FUSED_KERNELS = ("k_sg_runs32<baseline>", "k_sg_mask_span16<baseline>", "k_sg_mask<baseline>")
EXIT_GATHER_FAILED = 3


def parse_args() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preheat-steps", type=int, default=150,
                    help="untimed passes queued between the warm-up steps and the timed region: ~100 ms of load, the time "
                         "the GPU takes to reach its sustained clock from idle (profiles/r03_clock_ramp_probe.txt); 0 = none")
    ap.add_argument("--preset", default="v1725")
    ap.add_argument("--records", type=int, default=1_250_000, help="records per GPU (x800 = 1e9 samples)")
    ap.add_argument("--cpu-records", type=int, default=125_000, help="records in the CPU baseline slice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--threshold", type=float, default=10.0, help="hit threshold (reference default 10.0)")
    ap.add_argument("--no-features", action="store_true", help="skip the untimed feature / filter / end-to-end extras")
    ap.add_argument("--two-sessions", action="store_true",
                    help="also measure two sessions queuing passes on this GPU at the same time (informational; kernels of "
                         "the two streams overlap, so do not combine with a profiler run whose averages are to be compared)")
    ap.add_argument("--c4-records", type=int, default=333_334,
                    help="N > 1 only: records per GPU of the untimed config-4 leg (256-ch VX2730, x1500 = 5e8 samples); 0 = skip")
    ap.add_argument("--master-port", type=int, default=int(os.environ.get("WFA_BENCH_PORT", "29613")))
    return ap.parse_args()


def launch_ranks(args: argparse.Namespace) -> int:
    """`python bench.py --gpus N` without a launcher: start the ranks as a fresh child process tree (this process has not
    loaded libwfa_hip / torch and never will), relay rank 0's line, return the children's status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = next((ln for ln in reversed(proc.stdout.splitlines()) if ln.startswith("{")), None)
    if line is not None:
        print(line, flush=True)
    else:
        sys.stderr.write(proc.stdout)
    return proc.returncode if (line is not None or proc.returncode) else 1


def cpu_baseline(records, pool, n_records: int) -> dict:
    """Time the oracle (port of the reference loops) on the first n_records of the chunk: per-record scipy savgol_filter
    batches on a thread pool of os.cpu_count() workers (the reference's WavePoolFilteredPlugin default, max_workers=None
    -> all cores, cpu/records.py:405-430), then the single-threaded dense float64 matrix + per-hit loop of
    ThresholdHitPlugin (no parallelism there in the reference)."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from oracle import wfa_oracle as O
    from waveformanalysis_amd import synth

    L = int(records["event_length"][0])
    rec = records[:n_records].copy()
    sub = pool[: n_records * L]
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    rec["baseline"] = O.baseline_mean(sub.reshape(-1, L), 0, synth.BASELINE_SAMPLES)
    filt = np.zeros(len(sub), dtype=np.float32)
    batch = 1024  # records per task (the reference batches per channel x batch_size)
    bounds = [(lo, min(lo + batch, n_records)) for lo in range(0, n_records, batch)]

    def work(b):
        lo, hi = b
        part = rec[lo:hi].copy()
        part["wave_offset"] -= lo * L
        filt[lo * L : hi * L] = O.filter_wave_pool(part, sub[lo * L : hi * L])

    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, bounds))
    t_filter = time.perf_counter() - t0
    t1 = time.perf_counter()
    hits = O.threshold_hits_chunked(rec, filt, chunk=2048)  # dense f64 matrix + per-hit loop
    t_hits = time.perf_counter() - t1
    dt = t_filter + t_hits
    return {
        "value": round(n_records * L / dt / 1e9, 6),
        "unit": "Gsamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n_records} records x {L} samples of the same chunk ({n_records * L:.3g} samples): baseline mean "
                  f"+ per-record scipy savgol_filter on {cores} threads ({t_filter:.1f} s) + reference dense-matrix / "
                  f"per-hit loop on 1 thread ({t_hits:.1f} s)",
        "_hits": hits,
    }


def csrc_digest(repo: str = REPO) -> str:
    """First 16 hex digits of the sha256 over the kernel sources (csrc/*.hip, *.hpp, include/wfa_hip.h, in name order): what
    a PMC capture in profiles/hbm_traffic.json was taken from.  (The GPU box has no .git to ask.)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(repo, "waveformanalysis_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(repo, "waveformanalysis_amd", "csrc", "*.hpp")) +
                   [os.path.join(repo, "include", "wfa_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def traffic_entry(kname: str, preset: str, n_records: int, L: int):
    """HBM bytes per launch of `kname` on exactly this workload, from the PMC capture under profiles/ -- or None when there
    is none, or when the kernel sources have changed since it was taken (the entry's `csrc_sha16`)."""
    path = os.path.join(REPO, "profiles", "hbm_traffic.json")
    try:
        table = json.load(open(path))
    except Exception:
        return None, None
    ent = table.get(f"{kname}|{preset}|{n_records}|{L}")
    if not isinstance(ent, dict):
        return None, None
    src = {k: ent.get(k) for k in ("fetch_bytes", "write_bytes", "commit", "csrc_sha16", "source")}
    now = csrc_digest()
    if ent.get("csrc_sha16") != now:
        src["stale"] = f"kernel sources are {now} now: capture again (tools/collect_hbm_traffic.sh)"
        return None, src
    return ent.get("bytes"), src


def row_digest(rows) -> tuple[int, int]:
    """(count, 64-bit checksum) of a table of packed rows: the sum, modulo 2^64, of every row's 8-byte words (the tail of a
    row zero-extended) each multiplied by an odd constant of its position in the row -- rows may arrive in any order."""
    import numpy as np

    n, width = len(rows), rows.dtype.itemsize
    if n == 0:
        return 0, 0
    words = (width + 7) // 8
    buf = np.zeros((n, words * 8), dtype=np.uint8)
    buf[:, :width] = np.frombuffer(rows.tobytes(), dtype=np.uint8).reshape(n, width)
    mult = (np.arange(words, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    with np.errstate(over="ignore"):
        total = (buf.view(np.uint64) * mult[None, :]).sum(dtype=np.uint64)
    return n, int(total)


def config4_leg(args, DeviceSession, dist, rank: int, world: int, device_id: int) -> dict:
    """BASELINE config 4 under N > 1, outside the timed region: a 256-channel VX2730 run dealt to the ranks by channel,
    every rank its fused pass on the padded streaming route, all rows gathered over RCCL to rank 0, the gathered table
    VERIFIED there against every rank's own digest of what it sent (count + checksum, exchanged over the gloo control
    plane), then grouped from the device buffer (event_grouping.py:286-471)."""
    import numpy as np
    import torch

    from waveformanalysis_amd import synth
    from waveformanalysis_amd.dtypes import THRESHOLD_HIT_DTYPE

    out: dict = {"preset": "vx2730", "channels": 256, "records_per_gpu": int(args.c4_records)}
    rec4, pool4 = synth.make_run(args.c4_records, "vx2730", cfg=400 + rank)
    # rank r owns the channels {c : c mod world == r} of the 8 x 32 = 256: every record of this rank lands on one of them
    flat = rec4["board"].astype(np.int64) * 32 + rec4["channel"]
    flat = (flat // world) * world % 256 + rank if world <= 256 else flat
    rec4["board"], rec4["channel"] = flat // 32, flat % 32
    rec4["baseline"] = np.nan
    out["samples_per_gpu"] = int(pool4.size)
    s4 = DeviceSession(device_id)
    try:
        s4.upload_pool(pool4)
        s4.upload_records(rec4, args.threshold)
        s4.set_sg_plan(11, 2)
        n4 = s4.fused_baseline_filter_hits((0, synth.BASELINE_SAMPLES), 2, 2, download=False)
        mine = s4._fill_hits(n4)
        cnt, chk = row_digest(mine)
        digests = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(digests, torch.tensor([cnt, chk - (1 << 64) if chk >= (1 << 63) else chk], dtype=torch.int64))
        uid = [DeviceSession.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        s4.rccl_init(rank, world, uid[0])
        s4.rccl_gather_rows(None, n4, THRESHOLD_HIT_DTYPE, root=0, download=False)  # connection setup
        dist.barrier()
        t1 = time.perf_counter()
        counts, _none = s4.rccl_gather_rows(None, n4, THRESHOLD_HIT_DTYPE, root=0, download=False)
        out["gather_ms"] = round((time.perf_counter() - t1) * 1e3, 3)
        out["hits_total"] = int(counts.sum())
        _counts, table = s4.rccl_gather_rows(None, n4, THRESHOLD_HIT_DTYPE, root=0, download=True)  # untimed: the check
        if rank == 0:
            ok, at = True, 0
            for r in range(world):
                want_n, want_chk = int(digests[r][0]), int(digests[r][1]) % (1 << 64)
                got_n, got_chk = row_digest(table[at:at + int(counts[r])])
                ok = ok and got_n == want_n == int(counts[r]) and got_chk == want_chk
                at += int(counts[r])
            out["verified"] = bool(ok and at == len(table))
            s4.hit_rows_source("gather")
            t2 = time.perf_counter()
            flat_ev = s4.group_hit_windows_resident(out["hits_total"], 100.0)
            out["group_hit_windows_ms"] = round((time.perf_counter() - t2) * 1e3, 3)
            out["events"] = int(len(flat_ev["event_start"]) - 1)
            boards = np.unique(table["board"].astype(np.int64) * 32 + table["channel"])
            out["channels_seen"] = int(len(boards))
        out["ok"] = True
    finally:
        s4.close()
    return out


def rank_main(args: argparse.Namespace) -> int:
    # stdout carries exactly one JSON line: library banners (gloo, RCCL) are sent to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np

    from waveformanalysis_amd import _lib as _wfa_lib
    from waveformanalysis_amd import synth

    stub = os.environ.get("WFA_BENCH_STUB")  # tests only: a device stand-in, so that the launch / rendezvous / gather /
    if stub:                                 # exit-status logic of this file can be rehearsed on CPU (tests/bench_stub.py)
        import importlib

        DeviceSession = importlib.import_module(stub).Session
    else:
        _wfa_lib.load()  # bind /opt/rocm's HIP + RCCL before torch (which bundles its own copies) is imported
        from waveformanalysis_amd.device import DeviceSession

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        # torch.distributed is control plane only (rendezvous, barrier, max over ranks, RCCL id broadcast): gloo, so
        # torch never opens the GPU next to libwfa_hip's own HIP/RCCL runtime.
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group(backend="gloo")
    device_id = 0 if os.environ.get("WFA_BENCH_SHARE_GPU") else local_rank  # rehearsal on a 1-GPU box
    n_gpus = world

    # ---- synthetic chunk of this rank (channel shard = its own seed) ---------------------------------
    t0 = time.perf_counter()
    records, pool = synth.make_run(args.records, args.preset, cfg=100 + rank)
    if world > 1:
        # channel sharding: rank r holds hardware channels [r C / N, (r + 1) C / N) of the preset's C channels,
        # so the gathered hit table is one run over all of them for event grouping
        n_boards, ch_per_board = synth.PRESETS[args.preset][2], synth.PRESETS[args.preset][3]
        total_ch = n_boards * ch_per_board
        per_rank = max(1, total_ch // world)
        flat = (records["board"].astype(np.int64) * ch_per_board + records["channel"]) % per_rank + rank * per_rank
        flat %= total_ch
        records["board"], records["channel"] = flat // ch_per_board, flat % ch_per_board
    L = int(records["event_length"][0]) if len(records) else 0
    n_samples = int(pool.size)
    gen_s = time.perf_counter() - t0

    sess = DeviceSession(device_id)
    t0 = time.perf_counter()
    sess.upload_pool(pool)
    h2d_rate = sess.last_h2d_rate()
    rec_in = records.copy()
    rec_in["baseline"] = np.nan  # the fused pass estimates it
    sess.upload_records(rec_in, args.threshold)
    sess.set_sg_plan(11, 2)
    h2d_s = time.perf_counter() - t0

    def step() -> int:
        return sess.fused_baseline_filter_hits((0, synth.BASELINE_SAMPLES), 2, 2, download=False)

    def step_enqueue() -> None:
        # the same pass queued without a host round trip (wfa_hits_enqueue, the entry point the chunk-stream plugin
        # drives): consecutive passes run back to back on the device
        sess.hits_enqueue(_wfa_lib.SRC_SG_FUSED, (0, synth.BASELINE_SAMPLES), 2, 2)

    def sync_all() -> None:
        sess.sync()  # every launch of this rank is on the session's stream
        if dist is not None:
            dist.barrier()

    # The warm-up steps are the first GPU work after the host-side setup; their time is reported as `first_steps` (cold:
    # first touches, and a GPU that has idled through synthetic-data generation at a low clock).
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.warmup):
        step()
    sync_all()
    first_ms = (time.perf_counter() - t0) / max(args.warmup, 1) * 1e3 if args.warmup > 0 else None
    # Clock ramp: from idle the GPU needs ~30-50 ms of load to reach its sustained clock -- a 0.7 ms step is 35 % slower
    # in the first 20 steps after an idle half second than from the 40th on (tools/ramp_probe.py,
    # profiles/r03_clock_ramp_probe.txt).  A handful of warm-up steps is 3 ms of load, so the passes below bring the GPU
    # to the state a streaming run is in; they are the same enqueued pass as the timed ones, untimed.
    for _ in range(max(args.preheat_steps, 0)):
        step_enqueue()
    if args.preheat_steps > 0:
        sess.hits_wait()
    # timed region: HIP events only around the dominant (streaming) kernel -- the roofline figure needs its live
    # duration; event pairs around the small follow-up kernels of a pass cost about what the gaps between them do
    sess.profile(2)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_enqueue()
    sync_all()
    elapsed = time.perf_counter() - t0
    n_hits = sess.hits_wait()  # row count of the last pass (already complete)
    prof = sess.profile_report()
    # per-kernel breakdown of a pass: three more passes with every launch timed, outside the timed region
    sess.profile(1)
    for _ in range(3):
        step()
    sess.sync()
    prof_all = sess.profile_report()
    sess.profile(False)

    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- event-grouping exchange (RCCL over xGMI) + grouping on the root, outside the timed region ----------------
    gather = None
    gather_ok = True
    if dist is not None:
        import threading

        from waveformanalysis_amd.dtypes import THRESHOLD_HIT_DTYPE

        box: dict = {}

        def exchange() -> None:
            try:
                uid = [DeviceSession.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                sess.rccl_init(rank, world, uid[0])
                sess.rccl_gather_rows(None, n_hits, THRESHOLD_HIT_DTYPE, root=0, download=False)  # connection setup
                dist.barrier()
                t1 = time.perf_counter()
                counts, _none = sess.rccl_gather_rows(None, n_hits, THRESHOLD_HIT_DTYPE, root=0, download=False)
                box["ms"] = (time.perf_counter() - t1) * 1e3
                box["total"] = int(counts.sum())
                if rank == 0:  # grouping straight from the gathered device buffer
                    sess.hit_rows_source("gather")
                    t2 = time.perf_counter()
                    flat = sess.group_hit_windows_resident(box["total"], 100.0)
                    box["group_ms"] = (time.perf_counter() - t2) * 1e3
                    box["events"] = int(len(flat["event_start"]) - 1)
                    sess.hit_rows_source("hits")
                if args.c4_records > 0:
                    box["c4"] = config4_leg(args, DeviceSession, dist, rank, world, device_id)
                box["ok"] = True
            except Exception as exc:  # noqa: BLE001
                box["note"] = f"{type(exc).__name__}: {exc}"

        th = threading.Thread(target=exchange, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("WFA_BENCH_GATHER_TIMEOUT_S", "120")))
        hung = th.is_alive()
        gather_ok = bool(box.get("ok")) and not hung
        gather = {"ok": gather_ok, "hung": hung, "ms": box.get("ms"), "hits_total": box.get("total"),
                  "group_hit_windows_ms": box.get("group_ms"), "events": box.get("events"), "note": box.get("note"),
                  "time_window_ns": 100.0, "rows": "device-resident on rank 0 (no host copy)"}
        if box.get("c4") is not None:
            gather["config4"] = box["c4"]
            if rank == 0 and not box["c4"].get("verified", False):
                gather_ok = False
                gather["ok"] = False
                gather["note"] = (gather.get("note") or "") + " config-4 table does not match the ranks' digests"
        if not hung:
            # every rank learns whether any rank failed, so that all exit with the same status
            import torch

            flag = torch.tensor([0 if gather_ok else 1], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            gather_ok = gather_ok and int(flag.item()) == 0
            gather["ok"] = gather_ok

    # ---- the other kernels of the path on the same chunk, timed once each (not part of the metric) ----------------
    extra_ms, c3, e2e, two = {}, None, None, None
    if rank == 0 and not args.no_features and gather_ok:
        from waveformanalysis_amd import _lib as L_
        from waveformanalysis_amd.hit_merge import compute_cluster_rows, compute_merged_rows

        hit_rows = sess._fill_hits(n_hits)
        for timed in (False, True):  # first round allocates the scratch buffers, second round is reported
            if timed and args.preheat_steps > 0:  # the host work above left the GPU idle: same clock state as the timed region
                sess.profile(False)
                for _ in range(args.preheat_steps):
                    step_enqueue()
                sess.hits_wait()
            sess.profile(timed)
            sess.basic_features(L_.SRC_RAW)
            sess.width_integral(L_.SRC_RAW, dt=4.0)
            sess.features_both(dt=4.0, download=False)  # both tables from one read of the pool (config 3)
            sess.savgol(download=False)
            # find_peaks hit detector on the filtered pool that savgol just left resident (reference defaults)
            n_peaks = len(sess.find_peaks(L_.SRC_F32))
            # hit-table stages on this rank's threshold hits, straight from the rows the pass left on the device: merge
            # (gap 20 ns), then event grouping (100 ns)
            sess.hit_rows_source("hits")
            _order, offset = sess.hit_merge_clusters_resident(n_hits, 20.0, 10000.0)
            flat = sess.group_hit_windows_resident(n_hits, 100.0)
            clusters = compute_cluster_rows(sess, hit_rows, 20.0, 10000.0, None, "bench")
            merged = compute_merged_rows(sess, hit_rows, clusters, None, "bench")
            # records builder: global order of the records (already sorted: the sort still runs all passes)
            sess.records_sort_order(records["timestamp"], records["pid"], records["board"], records["channel"])
        rep = sess.profile_report()
        extra_ms = {k: round(v[0] / max(v[1], 1), 4) for k, v in rep.items()}
        extra_ms["_counts"] = {"peaks": int(n_peaks), "merged_hits": int(len(merged)), "clusters": int(len(offset) - 1),
                               "events": int(len(flat["event_start"]) - 1)}
        sess.profile(False)
        # BASELINE config 3: records -> hits -> BasicFeatures(area, width) on the 1e9-sample chunk
        pass_ms = sum(v[0] / max(v[1], 1) for v in prof_all.values())
        c3 = {"hits_pass_ms": round(pass_ms, 4),
              "k_basic_features_ms": extra_ms.get("k_basic_features_leaf", extra_ms.get("k_basic_features")),
              "k_width_integral_ms": extra_ms.get("k_width_integral_leaf", extra_ms.get("k_width_integral"))}
        c3["k_features_both_ms"] = extra_ms.get("k_features_both_leaf")
        if c3["k_features_both_ms"] is not None:
            # records -> hits -> both feature tables: the fused pass + ONE feature kernel that reads the pool once
            c3["total_ms"] = round(pass_ms + c3["k_features_both_ms"], 4)
            c3["Gsamples_per_s"] = round(n_samples / c3["total_ms"] / 1e6, 1)
            c3["algorithmic_bytes"] = int(2 * 2 * n_samples + (2 * 29 + 36 + 52) * len(records) + 60 * n_hits)
            c3["frac"] = round(c3["algorithmic_bytes"] / (c3["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            c3["note"] = "two reads of the pool (hit pass, feature kernel); frac on 2 x 2 B/sample + record columns + rows"
        elif c3["k_basic_features_ms"] is not None and c3["k_width_integral_ms"] is not None:
            c3["total_ms"] = round(pass_ms + c3["k_basic_features_ms"] + c3["k_width_integral_ms"], 4)
            c3["Gsamples_per_s"] = round(n_samples / c3["total_ms"] / 1e6, 1)
        # stand-alone roofline of the feature kernels: one read of the samples + the record columns + their rows
        for key, row_bytes in (("k_basic_features", 36), ("k_width_integral", 52), ("k_features_both", 36 + 52)):
            if c3.get(key + "_ms") is not None:
                b = 2 * n_samples + (29 + row_bytes) * len(records)
                c3[key + "_frac"] = round(b / (c3[key + "_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # end to end through the plugin boundary: host arrays in, structured rows out (H2D + kernels + D2H)
        try:
            from waveformanalysis_amd.plugin_api import SimpleContext
            from waveformanalysis_amd.plugins import HipThresholdHitPlugin

            ctx = SimpleContext({"wave_source": "records", "use_filtered": True, "fuse_filter": True,
                                 "fuse_baseline": (0, synth.BASELINE_SAMPLES), "threshold": args.threshold},
                                {"records": rec_in, "wave_pool": pool})
            ctx.wfa_device_pool = None
            plugin = HipThresholdHitPlugin()
            t1 = time.perf_counter()
            rows = plugin.compute(ctx, "bench")
            first = time.perf_counter() - t1
            # second call on the same thread's session (device buffers and the pinned ring exist); the pool is another
            # array object, so everything is uploaded again
            ctx2 = SimpleContext(dict(ctx.config), {"records": rec_in, "wave_pool": pool.copy()})
            ctx2.wfa_device_pool = None
            t1 = time.perf_counter()
            rows = plugin.compute(ctx2, "bench")
            dt = time.perf_counter() - t1
            e2e = {"value": round(n_samples / dt / 1e9, 3), "unit": "Gsamples/s", "seconds": round(dt, 4),
                   "first_call_seconds": round(first, 4),
                   "rows": int(len(rows)), "what": "HipThresholdHitPlugin.compute: H2D of pool + packed record rows (pinned "
                   "staging ring), records unpacked on the device, fused pass, D2H of the hit rows; first_call_seconds "
                   "includes creating the session, its device buffers and the pinned ring"}
        except Exception as exc:  # noqa: BLE001
            e2e = {"error": f"{type(exc).__name__}: {exc}"}
        # two sessions (= two HIP streams) on this GPU, each queuing passes over its own resident copy of the chunk: the
        # state the chunk-stream plugin keeps a card in (two sessions per device, streaming.py).  Informational: the
        # headline and the roofline figure above are one session's.
        try:
            if not args.two_sessions:
                raise LookupError("not requested")
            import threading

            other = DeviceSession(device_id)
            other.upload_pool(pool)
            other.upload_records(rec_in, args.threshold)
            other.set_sg_plan(11, 2)
            pair = [sess, other]

            def queue_passes(s, k):
                for _ in range(k):
                    s.hits_enqueue(_wfa_lib.SRC_SG_FUSED, (0, synth.BASELINE_SAMPLES), 2, 2)

            def both(k):
                th = [threading.Thread(target=queue_passes, args=(s, k)) for s in pair]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                return [s.hits_wait() for s in pair]

            both(max(args.preheat_steps, 20))
            k2 = max(args.steps, 100)
            t1 = time.perf_counter()
            counts = both(k2)
            dt2 = time.perf_counter() - t1
            two = {"ms_per_pass": round(dt2 / (2 * k2) * 1e3, 4), "value": round(2 * k2 * n_samples / dt2 / 1e9, 1),
                   "unit": "Gsamples/s", "passes": 2 * k2, "same_rows": bool(counts[0] == counts[1] == n_hits),
                   "what": "two sessions on one GPU queuing passes at the same time, each over its own resident copy of the chunk"}
            other.close()
        except LookupError:
            two = None
        except Exception as exc:  # noqa: BLE001
            two = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_samples * n_gpus / (elapsed / args.steps) / 1e9
        kname = next((k for k in FUSED_KERNELS if k in prof), None) or max(prof, key=lambda k: prof[k][0])
        k_ms, k_n = prof.get(kname, (0.0, 0))
        k_avg_s = (k_ms / k_n) * 1e-3 if k_n else float("nan")
        algo_bytes = 2 * n_samples + 29 * len(records) + 60 * n_hits
        achieved = algo_bytes / k_avg_s / 1e9 if k_n else float("nan")
        traffic, traffic_src = traffic_entry(kname, args.preset, len(records), L)
        out = {
            "metric": "Gsamples/s baseline+filter+hitfind",
            "value": round(value, 3),
            "unit": "Gsamples/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16 samples; int32 exact-rational SG + f64 hit windows",
            "data": "synthetic",
            "config": {
                "workload": f"{args.preset} {synth.PRESETS[args.preset][2] * synth.PRESETS[args.preset][3]}-ch synthetic chunk, {len(records)} records x {L} samples "
                            f"= {n_samples:.4g} samples per GPU; fused baseline(first 40) + SG(11,2) + "
                            "threshold hits (thr 10, ext 2/2), hit rows left on device",
                "samples_per_gpu": n_samples,
                "records_per_gpu": len(records),
                "hits_per_gpu": int(n_hits),
                "parallelism": f"channel-sharded x{n_gpus} (each rank its slice of the preset's channels), no data-path "
                               "collective; hit rows gathered over RCCL for event grouping",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": kname,
                "kernel_avg_ms": round(k_avg_s * 1e3, 4),
                "algorithmic_bytes": algo_bytes,
                "frac_pass": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            },
            "kernels_ms": {**{k: round(v[0] / max(v[1], 1), 4) for k, v in prof_all.items()},
                           **{k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}},
            "kernels_ms_note": f"{kname}: HIP events inside the timed region; the others: 3 extra passes after it",
            "clock_ramp": {"preheat_steps": int(args.preheat_steps),
                           "first_steps_ms_per_step": None if first_ms is None else round(first_ms, 4),
                           "note": "first_steps = the warm-up steps, the first GPU work after setup (cold clocks, first "
                                   "touches, one host round trip per step); preheat_steps untimed passes then bring the GPU to "
                                   "its sustained clock before the timed region (profiles/r03_clock_ramp_probe.txt)"},
            "other_kernels_ms": extra_ms,
            "setup": {"generate_s": round(gen_s, 2), "h2d_s": round(h2d_s, 3), "h2d_GBps": round(h2d_rate, 2),
                      "h2d_note": "wave_pool through two pinned 32-MiB staging buffers"},
        }
        if c3:
            out["config3"] = c3
        if e2e:
            out["end_to_end"] = e2e
        if two:
            out["two_sessions"] = two
        if gather is not None:
            out["gather_ok"] = gather_ok
            out["gather"] = gather
            if gather.get("ms") is not None:
                out["gather_ms"] = round(gather["ms"], 3)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a 1-GPU-run figure (rank 0, N = 1)
            n_cpu = min(args.cpu_records, len(records))
            cb = cpu_baseline(records, pool, n_cpu)
            cpu_hits = cb.pop("_hits")
            out["cpu_baseline"] = cb
            # parity on the slice: rows of the first n_cpu records, integer fields bit-exact
            n_chk = step()
            gpu_hits = sess._fill_hits(n_chk)
            sel = gpu_hits[gpu_hits["record_id"] < n_cpu]
            int_ok = len(sel) == len(cpu_hits) and all(
                np.array_equal(sel[f], cpu_hits[f]) for f in sel.dtype.names if sel.dtype[f].kind in "iu")
            flt = max((float(np.max(np.abs(sel[f] - cpu_hits[f]) / np.maximum(np.abs(cpu_hits[f]), 1e-30)))
                       for f in sel.dtype.names if sel.dtype[f].kind == "f"), default=0.0) if int_ok and len(sel) else None
            out["parity"] = {"records_checked": n_cpu, "hits_checked": int(len(cpu_hits)),
                             "int_fields_bit_exact": bool(int_ok), "max_rel_err_float_fields": flt}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if not gather_ok:
        # a failed or hung exchange is a finding: the line is out, every rank leaves with the same non-zero status
        # (no teardown: a stuck transport thread would block it)
        os._exit(EXIT_GATHER_FAILED)
    sess.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    if args.gpus != int(os.environ.get("WORLD_SIZE", "1")) and int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: the launcher decides",
              file=sys.stderr)
    return rank_main(args)


if __name__ == "__main__":
    sys.exit(main())
