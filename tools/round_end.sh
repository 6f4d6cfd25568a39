# full GPU suite, smoke, default bench, rocprofv3 kernel stats of the bench, VX2730 bench, instruction-rate microbenchmark
# -- outputs under gpurun_out/ (copy what is to be kept into profiles/ with the round's prefix)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2> gpurun_out/rocprof.err
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
rm -rf gpurun_out/prof
head -14 gpurun_out/kernel_stats.csv
timeout -k 10 600 python bench.py --preset vx2730 --records 666000 --cpu-records 20000 --no-features > gpurun_out/bench_vx2730.json
timeout -k 10 600 python bench.py --no-cpu-baseline --two-sessions > gpurun_out/bench_two_sessions.json 2> gpurun_out/bench_vx2730.err
cat gpurun_out/bench_vx2730.json
if [ -x tools/op_rates ]; then timeout -k 10 120 tools/op_rates > gpurun_out/op_rates.txt 2>&1; tail -20 gpurun_out/op_rates.txt; fi
