#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for n in 2 8 16; do
  for o in "19=1,9=32" "19=1,9=28" "19=1,9=36" "19=1,9=32,7=$((4+256))" "19=1,9=32,7=$((8+256))" "19=1,9=32,7=$((16+256))" "19=1,9=32,7=$((6+4*256))" "19=1,9=32,7=$((6+8*256))" "19=1,9=32,7=$((12+8*256))"; do
    echo -n "n=${n}Mi opts '$o': "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 5 --rays $((n<<20)) --opts "$o" 2>&1 | tail -1 | cut -d: -f2 | cut -c1-34
  done
done
