# A/B of two builds of the library inside one GPU-box call: gpurun_ab_old.so vs gpurun_ab_new.so (alternating).
# AB_KEYS: space-separated keys of the bench line's kernels_ms / other_kernels_ms to print (default: the pass kernels).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in old new old new old new; do
  cp gpurun_ab_$v.so waveformanalysis_amd/libwfa_hip.so
  echo -n "$v  "
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${AB_ARGS:---no-features} 2>/dev/null | python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); keys = os.environ.get('AB_KEYS', '').split()
src = {**d['kernels_ms'], **d.get('other_kernels_ms', {})}
print({k: src.get(k) for k in keys} if keys else d['kernels_ms'], d['ms_per_step'])"
done
cp gpurun_ab_new.so waveformanalysis_amd/libwfa_hip.so
