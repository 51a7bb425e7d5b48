#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for o in "" "12=32" "12=8" "5=2" "5=8" "13=60" "13=200" "13=400"; do
  echo -n "C4 64Mi opts '$o': "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 3 --opts "$o" 2>&1 | tail -1 | cut -d: -f2 | cut -c1-200
done
