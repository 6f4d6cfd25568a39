# SQ counters of the stream-probe kernels (tools/mfma_stream_probe), per kernel name, averaged over dispatches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmcP
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
         "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_WAVES_EQ_64" \
         "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TA_TCP_STATE_READ TCP_GATE_EN1 TCP_GATE_EN2 TA_BUSY TA_TA_BUSY" \
         "GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcP/p$i
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcP/p$i -- ./tools/mfma_stream_probe 2000 > /dev/null 2> gpurun_out/pmcP/p$i.err || { tail -3 gpurun_out/pmcP/p$i.err; }
  find gpurun_out/pmcP/p$i -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmcP/p$i.csv || true
  rm -rf gpurun_out/pmcP/p$i
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob("gpurun_out/pmcP/p*.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:70] + "|lds" + row.get("LDS_Block_Size", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    print("   ", {c: round(agg[k][c] / cnt[k][c] / 1e3, 1) for c in sorted(agg[k])}, "(thousands per launch)")
PY
