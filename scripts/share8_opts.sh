#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export RT_BENCH_QUICK=1 RT_BENCH_FORCE_GROUP=1 RT_BENCH_SHARE_OF=8 RT_BENCH_FRAMES=1
for o in "" "9=16" "9=21" "9=32" "7=$((4+256))" "7=$((8+256))" "7=$((12+256))" "6=64" "6=128"; do
  RT_BENCH_OPTS="$o" timeout -k 10 120 python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('share of 8, opts %-10s: %.3f ms/step   primary+seg0 %.3f  bounce %.3f  last+resolve %.3f' % ('$o', d['ms_per_step'], k['primary_shade_ms'], k['bounce_ms'], k['shade2_resolve_ms']))"
done
