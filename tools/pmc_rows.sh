set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  n=$(echo $c | cut -c1-12 | tr ' ' '_')
  rm -rf gpurun_out/pq
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-features > /dev/null 2> gpurun_out/pq.err || { tail -5 gpurun_out/pq.err; }
  find gpurun_out/pq -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pr_$n.csv || true
  find gpurun_out/pq -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/pr_trace.csv || true
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pr_SQ*.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "rows16" in k or "runs_to_desc" in k:
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in agg:
    print(k, {c: round(agg[k][c] / cnt[k][c] / 1e6, 2) for c in sorted(agg[k])}, "(millions per launch)")
dur = collections.defaultdict(list)
for row in csv.DictReader(open("gpurun_out/pr_trace.csv")):
    dur[row["Kernel_Name"].split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    if "rows16" in k or "runs" in k or "scan" in k: print(k, "avg us", round(sum(v) / len(v), 1), "n", len(v))
PY
