#!/bin/bash
# Bench frame with the staged ray queue (RT_OPT_RAY_QUEUE = option 19) against refill threshold, leaf-vote threshold and LDS stack cap.
# usage: scripts/sweep_queue.sh ["opts" ...]   each argument = one RT_BENCH_OPTS string; default set below
cd "$(dirname "$0")/.."
run() {
  RT_BENCH_OPTS="$1" RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-28s step %.3f ms  primary %.3f  bounce %.3f  shade2 %.3f  sha %s' % ('$1', d['ms_per_step'], k['primary_shade_ms'], k['bounce_ms'], k['shade2_resolve_ms'], d['frame_sha256'][:12]))" || { echo "$1 failed"; exit 1; }
}
if [ $# -gt 0 ]; then for o in "$@"; do run "$o"; done; exit 0; fi
run ""
for rf in 1 2 4 8 16; do run "19=1,9=12,7=$((rf + 12*256))"; done
for tth in 8 16 24; do run "19=1,9=12,7=$((4 + tth*256))"; done
run "19=1,9=20,7=$((4 + 12*256))"
run "19=1,9=10,7=$((4 + 12*256))"
