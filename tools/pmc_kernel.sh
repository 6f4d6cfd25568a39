# SQ / memory counters of the kernels of the fused pass (separate rocprofv3 passes, counters only), summed per kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_FLAT" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcK_$i
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcK_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-features > gpurun_out/pmcK_$i.json 2> gpurun_out/pmcK_$i.err || { tail -5 gpurun_out/pmcK_$i.err; }
  find gpurun_out/pmcK_$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmcK_$i.csv || true
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob("gpurun_out/pmcK_*.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in agg:
    if "runs32" in k or "rows_grp" in k or "span16" in k:
        print(k)
        for c in sorted(agg[k]): print(f"   {c:24s} {agg[k][c] / cnt[k][c]:16.1f}  (per launch, {cnt[k][c]} launches)")
PY
