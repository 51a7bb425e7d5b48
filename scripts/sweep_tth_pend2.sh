cd "$GRAFT_REPO_ROOT"
for tth in 8 12 20 28 36 44; do
  v=$((16 | (tth << 8)))
  RT_AMD_LIB=$PWD/raytracer_amd/csrc/libraytracer_amd_pend2.so RT_BENCH_OPTS="7=$v" RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('pend2 tth $tth: bounce %.3f  N=%.2f T=%.2f' % (k['bounce_ms'], d['roofline']['work_per_ray']['N'], d['roofline']['work_per_ray']['T']))"
done
