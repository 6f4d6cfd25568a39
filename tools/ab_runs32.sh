# A/B of the streaming kernel variants on one box: parity tests of the fused path first, then bench lines
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_padded.py tests/test_hip_fullsize.py -x -q -m gpu > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -3 gpurun_out/ab_tests.log
for v in ${AB_VARIANTS:-"WFA_RUNS_DBG=0" "WFA_RUNS_DBG=1" "WFA_RUNS_DBG=2" "WFA_RUNS_DBG=4" "WFA_RUNS_DBG=8" "WFA_RUNS_DBG=15" "WFA_DISABLE_RUNS32=1"}; do
  echo "== $v"
  env $v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-features > gpurun_out/ab_$(echo $v | tr ' =' '__').json 2> gpurun_out/ab_err.log || { tail -20 gpurun_out/ab_err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$(echo $v | tr ' =' '__').json"))
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_avg_ms"], d["roofline"]["frac"], d["config"]["hits_per_gpu"])
print({k: v for k, v in d["kernels_ms"].items()})
PY
done
