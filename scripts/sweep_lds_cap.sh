#!/bin/bash
# Bench frame against the LDS part of the ray-buffer kernel's stack (RT_OPT_LDS_STACK = option 9): what a smaller cap costs,
# i.e. how much LDS per block could go to something else (a staged ray queue) at the same occupancy.
cd "$(dirname "$0")/.."
for cap in ${CAPS:-8 10 12 14 16 20}; do
  RT_BENCH_OPTS="9=$cap" RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('cap $cap: step %.3f ms  primary %.3f  bounce %.3f  shade2 %.3f  sha %s' % (d['ms_per_step'], k['primary_shade_ms'], k['bounce_ms'], k['shade2_resolve_ms'], d['frame_sha256'][:12]))" || { echo "cap $cap failed"; exit 1; }
done
