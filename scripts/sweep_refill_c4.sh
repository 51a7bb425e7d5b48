#!/bin/bash
# C4 soup (64 Mi rays): refill threshold (idle lanes) x leaf-vote threshold of the ray-buffer kernel
cd "$GRAFT_REPO_ROOT"
for rf in 8 16 24 32; do for tth in 4 8 12 20; do
  echo -n "refill $rf tth $tth: "; timeout -k 10 120 python -u scripts/c4_prof.py --reps 3 --refill $((rf | (tth << 8))) 2>&1 | tail -1 | cut -d: -f2 | cut -c1-32
done; done
