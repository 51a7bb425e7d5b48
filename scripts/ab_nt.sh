#!/bin/bash
# nontemporal ray loads / hit stores for binned launches (StripScratch.stream_nt): bench frame, C4 at 64 Mi and 8 Mi rays, C5, host path
cd "$(dirname "$0")/.."
scripts/sweep_queue.sh ""
echo -n "C4 64Mi: "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 3 2>&1 | tail -1 | cut -d: -f2 | cut -c1-34
echo -n "C4  8Mi: "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 5 --rays 8388608 2>&1 | tail -1 | cut -d: -f2 | cut -c1-34
echo -n "C5: "; timeout -k 10 200 python -u scripts/c5_stages.py 2>&1 | tail -1 | cut -c1-200
