# SQ counters of one kernel (PMC_KERNEL, default the streaming kernel k_sg_runs32) under bench.py or PMC_SCRIPT, per unit
# (PMC_DIV units per launch; default 488281.25 = the 2048-sample tiles of the default chunk)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmcS
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES SQ_CYCLES" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcS/p$i
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcS/p$i -- python3 ${PMC_SCRIPT:-bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-features} ${PMC_BENCH_ARGS:-} > /dev/null 2> gpurun_out/pmcS/p$i.err || { tail -3 gpurun_out/pmcS/p$i.err; }
  find gpurun_out/pmcS/p$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmcS/p$i.csv || true
  rm -rf gpurun_out/pmcS/p$i
done
python3 - <<'PY'
import csv, glob, collections, os
sel = os.environ.get("PMC_KERNEL", "runs32")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob("gpurun_out/pmcS/p*.csv")):
    for row in csv.DictReader(open(f)):
        if sel not in row["Kernel_Name"]: continue
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
div = float(os.environ.get("PMC_DIV", "488281.25"))
for k in sorted(agg):
    print(k)
    print("   per unit:", {c: round(agg[k][c] / cnt[k][c] / div, 2) for c in sorted(agg[k])})
    print("   millions per launch:", {c: round(agg[k][c] / cnt[k][c] / 1e6, 2) for c in sorted(agg[k])})
PY
