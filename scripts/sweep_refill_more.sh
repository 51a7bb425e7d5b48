#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rf in 32 40 48 56; do
  echo -n "soup 64Mi refill $rf tth 8: "; timeout -k 10 120 python -u scripts/c4_prof.py --reps 3 --refill $((rf | (8 << 8))) 2>&1 | tail -1 | cut -d: -f2 | cut -c1-32
  echo -n "soup 8Mi  refill $rf tth 8: "; timeout -k 10 120 python -u scripts/c4_prof.py --rays 8388608 --reps 5 --refill $((rf | (8 << 8))) 2>&1 | tail -1 | cut -d: -f2 | cut -c1-32
done
for rf in 12 16 20 24 32; do
  v=$((rf | (12 << 8)))
  RT_BENCH_OPTS="7=$v" RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('bench refill $rf tth 12: bounce %.3f' % k['bounce_ms'])"
done
