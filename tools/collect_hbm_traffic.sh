# HBM traffic per launch of the fused pass's kernels: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE separately, counters
# only -- no trace domains), reduced to profiles-style JSON by tools/reduce_hbm_traffic.py
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-features > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
  find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmc_$c.csv
done
python tools/reduce_hbm_traffic.py gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv > gpurun_out/hbm_traffic.json
cat gpurun_out/hbm_traffic.json
