#!/bin/bash
# C4 soup (64 Mi and 8 Mi rays) and the C5 frame with and without the staged ray queue (option 19), LDS cap (9), thresholds (7)
cd "$(dirname "$0")/.."
c4() { echo -n "C4 rays $1 opts '$2': "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 3 --rays $1 --opts "$2" 2>&1 | tail -1 | cut -d: -f2 | cut -c1-34; }
for n in $((64<<20)) $((8<<20)); do
  c4 $n "" || exit 1
  c4 $n "19=1,9=12,7=2052" || exit 1
  c4 $n "19=1,9=20,7=2052" || exit 1
  c4 $n "19=1,9=12,7=2064" || exit 1
  c4 $n "19=1,9=24,7=2064" || exit 1
done
for o in "" "19=1,9=12,7=2052" "19=1,9=12,7=3076"; do echo -n "C5 opts '$o': "; RT_OPTS="$o" timeout -k 10 200 python -u scripts/c5_stages.py 2>&1 | tail -1; done
