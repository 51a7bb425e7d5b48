#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for st in 64 128 256 512; do
  RT_BENCH_OPTS="6=$st" RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('bench strip $st: bounce %.3f' % k['bounce_ms'])"
done
for st in 64 256; do echo -n "C5 strip $st: "; RT_OPTS="6=$st" timeout -k 10 200 python -u scripts/c5_stages.py 2>&1 | tail -1 | cut -c1-150; done
