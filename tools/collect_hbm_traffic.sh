# HBM traffic per launch of the kernels bench.py runs: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE separately, counters
# only -- no trace domains), reduced by tools/reduce_hbm_traffic.py into gpurun_out/hbm_traffic.json (copy to profiles/).
# usage (GPU box): COMMIT=<git sha> [PRESET=v1725 RECORDS=1250000 L=800] bash tools/collect_hbm_traffic.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PRESET=${PRESET:-v1725}; RECORDS=${RECORDS:-1250000}; L=${L:-800}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --preset $PRESET --records $RECORDS > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
  find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmc_${PRESET}_$c.csv
  rm -rf gpurun_out/pmc_$c
done
python tools/reduce_hbm_traffic.py gpurun_out/pmc_${PRESET}_FETCH_SIZE.csv gpurun_out/pmc_${PRESET}_WRITE_SIZE.csv $PRESET $RECORDS $L "${COMMIT:-unknown}" profiles/hbm_traffic.json > gpurun_out/hbm_traffic.json
python - <<'PY'
import json
t = json.load(open("gpurun_out/hbm_traffic.json"))
for k, v in t.items():
    if isinstance(v, dict): print(k, v["bytes"], "fetch", v["fetch_bytes"], "write", v["write_bytes"])
PY
