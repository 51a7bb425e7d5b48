#!/bin/bash
# Leaf size (rt_set_option 5 = RT_OPT_MAX_LEAF) on the bench frame: both traversal kernels.
cd "$(dirname "$0")/.."
for o in "5=1" "5=2" "5=3" "5=4" "5=6" "5=8"; do
  RT_BENCH_OPTS=$o RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; w=d['roofline']['work_per_ray']; p=k['primary_work_per_ray']
print('opts [$o]: step %.3f primary %.3f (N %.1f T %.1f) bounce %.3f (N %.1f T %.1f) nodes %d' % (d['ms_per_step'], k['primary_shade_ms'], p['N'], p['T'], k['bounce_ms'], w['N'], w['T'], d['config']['bvh']['nodes']))"
done
