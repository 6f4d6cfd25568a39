# full GPU suite, smoke, default bench, rocprofv3 kernel stats of the bench -- outputs under gpurun_out/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
tail -1 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
rm -rf gpurun_out/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2> gpurun_out/rocprof.err
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
head -12 gpurun_out/kernel_stats.csv
