# instruction / wait counters per launch for the kernels whose name matches PMC_MATCH (regex), from one bench.py run each
# usage (GPU box): PMC_MATCH='features_wave|basic_features|width_integral' PMC_ARGS='--steps 2 --warmup 1' bash tools/pmc_rows.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/pr_SQ*.csv gpurun_out/pr_trace.csv
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -c1-12 | tr ' ' '_')
  rm -rf gpurun_out/pq
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pq -- python3 bench.py ${PMC_ARGS:---steps 2 --warmup 1} --no-cpu-baseline > /dev/null 2> gpurun_out/pq.err || { tail -5 gpurun_out/pq.err; }
  find gpurun_out/pq -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pr_$n.csv || true
  find gpurun_out/pq -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/pr_trace.csv || true
done
python3 - <<'PY'
import csv, glob, collections, os, re
pat = re.compile(os.environ.get("PMC_MATCH", "rows|runs"))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pr_SQ*.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if pat.search(k):
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in agg:
    print(k, {c: round(agg[k][c] / cnt[k][c] / 1e6, 2) for c in sorted(agg[k])}, "(millions per launch)")
dur = collections.defaultdict(list)
for row in csv.DictReader(open("gpurun_out/pr_trace.csv")):
    dur[row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    if pat.search(k): print(k, "avg us", round(sum(v) / len(v), 1), "n", len(v))
PY
