cp waveformanalysis_amd/libwfa_hip.so /tmp/new.so
for v in new runs128 runs512 new runs128 runs512; do
  if [ $v = new ]; then cp /tmp/new.so waveformanalysis_amd/libwfa_hip.so; else cp tools/libwfa_hip_$v.so waveformanalysis_amd/libwfa_hip.so; fi
  echo -n "$v  "
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-features 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_ms']['k_hit_runs'], d['ms_per_step'])"
done
cp /tmp/new.so waveformanalysis_amd/libwfa_hip.so
