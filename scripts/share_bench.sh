#!/bin/bash
# Predicted strong scaling as bench.py itself runs it: rank 0's share of a world of M ranks (RT_BENCH_SHARE_OF) on one GPU, with the
# pipelined exchange of that share through a one-rank RCCL group and F frames in flight.  ms per step of the share against
# (ms per step of the whole frame) / M.
cd "$(dirname "$0")/.."
export RT_BENCH_QUICK=1
run() {  # M F
  RT_BENCH_FORCE_GROUP=1 RT_BENCH_SHARE_OF=$1 RT_BENCH_FRAMES=$2 timeout -k 10 120 python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('share of $1, $2 frame(s) in flight: %.3f ms/step (host queues a step in %.3f ms)   single launches: primary+seg0 %.3f  bounce %.3f  last+resolve %.3f' % (d['ms_per_step'], d['host_enqueue_ms_per_step'], k['primary_shade_ms'], k['bounce_ms'], k['shade2_resolve_ms']))" || { echo "M=$1 F=$2 failed"; exit 1; }
}
echo -n "whole frame, no group: "; timeout -k 10 120 python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.3f ms/step' % d['ms_per_step'])"
for M in 1 2 4 8; do for F in ${FRAMES:-1 2 3 4}; do run $M $F; done; done
