#!/usr/bin/env python3
"""bench.py -- Gsamples/s of the fused baseline + Savitzky-Golay filter + threshold-hit pass.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--records R] [--preset v1725]

One *step* = one pass of the hot path over one resident chunk: the fused kernel
(baseline estimate over the first 40 samples, SG(11,2) filter evaluated on the fly, threshold
hits) + the hit-count scan + the gather into (record, start) order.  Inputs (wave_pool uint16,
records SoA) are resident in HBM before the timed region; hit rows stay on the device.

N > 1: launched by torch.distributed.run, one rank per GPU.  Records shard by channel with no
data-path collective, so every rank processes its own equally sized shard
("weak" scaling) and value = all samples / max-over-ranks time.  After the timed region the
ranks' hits are gathered to rank 0 over RCCL (the event-grouping exchange) and that time is
reported separately as gather_ms.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit..., plus
  roofline     achieved = (2*N + 29*R + 60*H) bytes / mean duration of the fused kernel, measured
               with HIP events on the kernel's own stream inside the timed region; peak 8000 GB/s
  cpu_baseline the oracle's literal reference loops (scipy savgol per record + per-hit python
               loop) timed on a bounded slice of the same chunk on this host, single thread
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from waveformanalysis_amd import synth  # noqa: E402
from waveformanalysis_amd.device import DeviceSession  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# dominant (streaming) kernel of the fused pass, by preference: uniform records on the run-event kernel, else the
# per-record mask kernels
FUSED_KERNELS = ("k_sg_runs32<baseline>", "k_sg_mask_span16<baseline>", "k_sg_mask<baseline>")


def cpu_baseline(records: np.ndarray, pool: np.ndarray, n_records: int) -> dict:
    """Time the oracle (port of the reference loops) on the first n_records of the chunk."""
    from oracle import wfa_oracle as O

    L = int(records["event_length"][0])
    rec = records[:n_records].copy()
    sub = pool[: n_records * L]
    t0 = time.perf_counter()
    rec["baseline"] = O.baseline_mean(sub.reshape(-1, L), 0, synth.BASELINE_SAMPLES)
    filt = O.filter_wave_pool(rec, sub)  # per-record scipy savgol_filter, as records.py:368-438
    hits = O.threshold_hits_chunked(rec, filt, chunk=2048)  # dense f64 matrix + per-hit loop
    dt = time.perf_counter() - t0
    return {
        "value": round(n_records * L / dt / 1e9, 6),
        "unit": "Gsamples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n_records} records x {L} samples of the same chunk "
                  f"({n_records * L:.3g} samples, {dt:.1f} s): baseline mean + per-record scipy "
                  "savgol_filter + reference per-hit loop",
        "_hits": hits,
    }


def main() -> None:
    # stdout carries exactly one JSON line: library banners (gloo, RCCL) are sent to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from waveformanalysis_amd import _lib as _wfa_lib

    _wfa_lib.load()  # bind /opt/rocm's HIP + RCCL before torch (which bundles its own copies) is imported
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preset", default="v1725")
    ap.add_argument("--records", type=int, default=1_250_000, help="records per GPU (x800 = 1e9 samples)")
    ap.add_argument("--cpu-records", type=int, default=125_000, help="records in the CPU baseline slice")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--threshold", type=float, default=10.0, help="hit threshold (reference default 10.0)")
    ap.add_argument("--no-features", action="store_true", help="skip the untimed feature / filter kernels")
    ap.add_argument("--grouping", action="store_true",
                    help="also run the event grouping of the (gathered) hit rows on rank 0 (untimed extra)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        # torch.distributed is control plane only (rendezvous, barrier, max over ranks, RCCL id
        # broadcast): gloo, so torch never opens the GPU next to libwfa_hip's own HIP/RCCL runtime.
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group(backend="gloo")
    device_id = 0 if os.environ.get("WFA_BENCH_SHARE_GPU") else local_rank  # rehearsal on a 1-GPU box
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}; using {n_gpus}", file=sys.stderr)

    # ---- synthetic chunk of this rank (channel shard = its own seed) ---------------------------------
    t0 = time.perf_counter()
    records, pool = synth.make_run(args.records, args.preset, cfg=100 + rank)
    L = int(records["event_length"][0]) if len(records) else 0
    n_samples = int(pool.size)
    gen_s = time.perf_counter() - t0

    sess = DeviceSession(device_id)
    t0 = time.perf_counter()
    sess.upload_pool(pool)
    rec_in = records.copy()
    rec_in["baseline"] = np.nan  # the fused pass estimates it
    sess.upload_records(rec_in, args.threshold)
    sess.set_sg_plan(11, 2)
    h2d_s = time.perf_counter() - t0

    def step() -> int:
        return sess.fused_baseline_filter_hits((0, synth.BASELINE_SAMPLES), 2, 2, download=False)

    def step_enqueue() -> None:
        # the same pass queued without a host round trip: consecutive passes run back to back on the device
        sess.hits_enqueue(_wfa_lib.SRC_SG_FUSED, (0, synth.BASELINE_SAMPLES), 2, 2)

    def sync_all() -> None:
        sess.sync()  # every launch of this rank is on the session's stream
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    # timed region: HIP events only around the dominant (streaming) kernel -- the roofline figure needs its live
    # duration; event pairs around the four small follow-up kernels of a pass cost about what the gaps between them do
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

    # ---- event-grouping exchange (RCCL over xGMI), outside the timed region ---------------------------
    gather_ms = None
    gather_note = None
    rows = None
    total_hits = n_hits
    if dist is not None:
        from waveformanalysis_amd.dtypes import THRESHOLD_HIT_DTYPE

        # The exchange is reported next to the metric, which does not depend on it: it runs on a watchdog thread so that
        # a transport that never comes up (the 8-GPU node is not available to the build sessions) cannot take the
        # measured line with it.
        import threading

        box: dict = {}

        def exchange() -> None:
            try:
                uid = [DeviceSession.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                sess.rccl_init(rank, world, uid[0])
                sess.rccl_gather_rows(None, n_hits, THRESHOLD_HIT_DTYPE, root=0)  # warm-up (connection setup)
                dist.barrier()
                t1 = time.perf_counter()
                counts, got = sess.rccl_gather_rows(None, n_hits, THRESHOLD_HIT_DTYPE, root=0)
                box["ms"] = (time.perf_counter() - t1) * 1e3
                box["total"] = int(counts.sum())
                box["rows"] = got
                if rank == 0:
                    assert got is not None and len(got) == box["total"]
            except Exception as exc:  # noqa: BLE001
                box["note"] = f"RCCL gather not run: {exc}"

        th = threading.Thread(target=exchange, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("WFA_BENCH_GATHER_TIMEOUT_S", "120")))
        if th.is_alive():
            gather_note = "RCCL gather did not finish within the watchdog time; metric line unaffected"
            hung = True
        else:
            hung = False
            gather_ms, gather_note = box.get("ms"), box.get("note")
            rows = box.get("rows")
            total_hits = box.get("total", n_hits)
    else:
        hung = False

    # ---- the other per-record kernels of the path, timed once each (not part of the metric) ----------
    extra_ms = {}
    if rank == 0 and not args.no_features and not hung:
        from waveformanalysis_amd import _lib as L_

        from waveformanalysis_amd.event_grouping import group_hit_windows_flat
        from waveformanalysis_amd.hit_merge import compute_cluster_rows, compute_merged_rows

        hit_rows = sess._fill_hits(n_hits)
        for timed in (False, True):  # first round allocates the scratch buffers, second round is reported
            sess.profile(timed)
            sess.basic_features(L_.SRC_RAW)
            sess.width_integral(L_.SRC_RAW, dt=4.0)
            sess.savgol(download=False)
            # find_peaks hit detector on the filtered pool that savgol just left resident (reference defaults)
            n_peaks = len(sess.find_peaks(L_.SRC_F32))
            # hit-table stages on this rank's threshold hits: merge (gap 20 ns), then event grouping (100 ns)
            clusters = compute_cluster_rows(sess, hit_rows, 20.0, 10000.0, None, "bench")
            merged = compute_merged_rows(sess, hit_rows, clusters, None, "bench")
            flat = group_hit_windows_flat(hit_rows, 100.0, session=sess)
            # records builder: global order of the records (already sorted: the sort still runs all passes)
            sess.records_sort_order(records["timestamp"], records["pid"], records["board"], records["channel"])
        extra_ms = {k: round(v[0] / max(v[1], 1), 4) for k, v in sess.profile_report().items()}
        extra_ms["_counts"] = {"peaks": int(n_peaks), "merged_hits": int(len(merged)),
                               "events": int(len(flat["event_start"]) - 1)}
        sess.profile(False)

    grouping = None
    if rank == 0 and args.grouping and not hung:
        from waveformanalysis_amd.event_grouping import group_hit_windows_flat

        all_rows = rows if (dist is not None and gather_ms is not None) else sess._fill_hits(n_hits)
        t0 = time.perf_counter()
        flat = group_hit_windows_flat(all_rows, 100.0)
        grouping = {"hits": int(len(all_rows)), "events": int(len(flat["event_start"]) - 1),
                    "host_ms": round((time.perf_counter() - t0) * 1e3, 1), "time_window_ns": 100.0}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_samples * n_gpus / (elapsed / args.steps) / 1e9
        kname = next((k for k in FUSED_KERNELS if k in prof), None) or max(prof, key=lambda k: prof[k][0])
        k_ms, k_n = prof.get(kname, (0.0, 0))
        k_avg_s = (k_ms / k_n) * 1e-3 if k_n else float("nan")
        algo_bytes = 2 * n_samples + 29 * len(records) + 60 * n_hits
        achieved = algo_bytes / k_avg_s / 1e9 if k_n else float("nan")
        traffic = None
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kname)
            except Exception:
                traffic = None
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
                "parallelism": f"channel-sharded x{n_gpus}, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": kname,
                "kernel_avg_ms": round(k_avg_s * 1e3, 4),
                "algorithmic_bytes": algo_bytes,
            },
            "kernels_ms": {**{k: round(v[0] / max(v[1], 1), 4) for k, v in prof_all.items()},
                           **{k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}},
            "kernels_ms_note": f"{kname}: HIP events inside the timed region; the others: 3 extra passes after it",
            "other_kernels_ms": extra_ms,
            "setup": {"generate_s": round(gen_s, 2), "h2d_s": round(h2d_s, 3),
                      "h2d_GBps": round((2 * n_samples) / h2d_s / 1e9, 2)},
        }
        if gather_ms is not None:
            out["gather_ms"] = round(gather_ms, 3)
            out["config"]["hits_total"] = total_hits
        if gather_note:
            out["gather_note"] = gather_note
        if grouping:
            out["event_grouping"] = grouping
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a 1-GPU-run figure (rank 0, N = 1)
            n_cpu = min(args.cpu_records, len(records))
            cb = cpu_baseline(records, pool, n_cpu)
            cpu_hits = cb.pop("_hits")
            out["cpu_baseline"] = cb
            # parity on the slice: rows of the first n_cpu records, integer fields bit-exact
            gpu_hits = sess._fill_hits(n_hits)
            sel = gpu_hits[gpu_hits["record_id"] < n_cpu]
            int_ok = len(sel) == len(cpu_hits) and all(
                np.array_equal(sel[f], cpu_hits[f]) for f in sel.dtype.names if sel.dtype[f].kind in "iu")
            flt = max((float(np.max(np.abs(sel[f] - cpu_hits[f]) / np.maximum(np.abs(cpu_hits[f]), 1e-30)))
                       for f in sel.dtype.names if sel.dtype[f].kind == "f"), default=0.0) if int_ok and len(sel) else None
            out["parity"] = {"records_checked": n_cpu, "hits_checked": int(len(cpu_hits)),
                             "int_fields_bit_exact": bool(int_ok), "max_rel_err_float_fields": flt}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if hung:
        os._exit(0)  # a stuck transport thread would block the teardown; the line is out
    sess.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
