#!/bin/bash
# RT_OPT_SAH_TRAV (option 13: SAH cost of an interior step in hundredths of a triangle test) on the soup (C4, C5) and the bench frame
cd "$(dirname "$0")/.."
for v in ${VALUES:-20 30 40 60 80 120}; do
  echo -n "C4 64Mi 13=$v: "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 3 --opts "13=$v" 2>&1 | tail -1 | cut -d: -f2 | cut -c1-120
  echo -n "C4  8Mi 13=$v: "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 3 --rays $((8<<20)) --opts "13=$v" 2>&1 | tail -1 | cut -d: -f2 | cut -c1-60
  echo -n "C5      13=$v: "; RT_OPTS="13=$v" timeout -k 10 200 python -u scripts/c5_stages.py 2>&1 | tail -1 | cut -c1-200
  scripts/sweep_queue.sh "13=$v"
done
