# instruction counters of the streaming kernel for a list of WFA_RUNS_DBG values
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for d in ${PMC_DBG:-0 15}; do
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_WR"; do
    n=$(echo $c | cut -c1-12 | tr ' ' '_')
    rm -rf gpurun_out/pq
    WFA_RUNS_DBG=$d timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-features > /dev/null 2> gpurun_out/pq.err || { tail -5 gpurun_out/pq.err; }
    find gpurun_out/pq -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pq_${d}_$n.csv || true
  done
done
python3 - <<'PY'
import csv, glob, collections, os
KSEL = os.environ.get("PMC_KERNEL", "runs32")
for d in sorted(set(f.split("_")[2] for f in glob.glob("gpurun_out/pq_*_*.csv"))):
    agg = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/pq_{d}_*.csv"):
        for row in csv.DictReader(open(f)):
            if ("runs32" in row["Kernel_Name"]) == (KSEL == "runs32") and (KSEL in row["Kernel_Name"]):
                agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
    print("DBG", d, {c: round(agg[c] / cnt[c] / float(os.environ.get("PMC_DIV", "488281.25")), 1) for c in sorted(agg)}, "(per unit)")
PY
