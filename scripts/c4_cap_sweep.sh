#!/bin/bash
# C4 soup: stack entries kept in LDS (RT_OPT_LDS_STACK, option 9) against the ray count, with and without the staged ray queue
cd "$(dirname "$0")/.."
for n in 2 4 8 16 32; do
  for o in "" "9=16" "9=21" "9=26" "9=32" "9=40" "19=1,9=16" "19=1,9=24" "19=1,9=32"; do
    echo -n "n=${n}Mi opts '$o': "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 5 --rays $((n<<20)) --opts "$o" 2>&1 | tail -1 | cut -d: -f2 | cut -c1-34
  done
done
