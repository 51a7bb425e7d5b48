#!/bin/bash
# Builder-parameter sweep: the bench frame (C3) and the C4 soup under rt_set_option pairs (5 = max leaf, 12 = SAH bins, 13 = SAH trav x100).
cd "$(dirname "$0")/.."
for o in "" "13=60" "13=30" "13=10" "5=2,13=60" "5=2,13=30" "5=8,13=60" "12=32,13=60"; do
  RT_BENCH_OPTS=$o RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; w=d['roofline']['work_per_ray']; p=k['primary_work_per_ray']
print('C3 opts [$o]: step %.3f primary %.3f (N %.1f T %.1f) bounce %.3f (N %.1f T %.1f) nodes %d build %.0f ms' % (d['ms_per_step'], k['primary_shade_ms'], p['N'], p['T'], k['bounce_ms'], w['N'], w['T'], d['config']['bvh']['nodes'], d['config']['bvh']['build_ms']))"
done
for o in "13=60" "13=30" "13=10" "5=2,13=30" "5=2,13=10"; do
  timeout -k 10 200 python -u scripts/c4_prof.py --rays 16777216 --reps 3 --opts "$o" 2>&1 | grep "C4 soup"
done
