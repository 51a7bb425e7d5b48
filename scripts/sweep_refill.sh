#!/bin/bash
# Scheduling sweep of the ray-buffer kernel on the bench frame: rt_set_option 7 (RT_OPT_REFILL) = leaf-vote threshold << 8 | idle lanes for a refill
cd "$(dirname "$0")/.."
for tth in 12 16 20 24 32; do for rf in 8 12 16 24; do
  v=$((tth * 256 + rf))
  RT_BENCH_OPTS="7=$v" RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('leaf vote $tth refill $rf: step %.3f bounce %.3f N %.2f sha %s' % (d['ms_per_step'], k['bounce_ms'], d['roofline']['work_per_ray']['N'], d['frame_sha256'][:12]))"
done; done
