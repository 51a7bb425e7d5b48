#!/bin/bash
# A/B of library builds (compiler scheduling strategies) beyond the bench frame: C4 at 64 Mi / 8 Mi rays, C5 stages, C2, C0
# usage: scripts/ab_sched.sh <tag> ...   ("base" = libraytracer_amd.so)
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=$PWD/raytracer_amd/csrc/libraytracer_amd.so; [ "$tag" != base ] && lib=$PWD/raytracer_amd/csrc/libraytracer_amd_$tag.so
  export RT_AMD_LIB=$lib
  echo -n "$tag C4 64Mi: "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 3 2>&1 | tail -1 | cut -d: -f2 | cut -c1-34
  echo -n "$tag C4  8Mi: "; timeout -k 10 150 python -u scripts/c4_prof.py --reps 5 --rays 8388608 2>&1 | tail -1 | cut -d: -f2 | cut -c1-34
  echo -n "$tag C5: "; timeout -k 10 200 python -u scripts/c5_stages.py 2>&1 | tail -1 | cut -c1-200
  echo -n "$tag C2: "; timeout -k 10 100 python -u scripts/c2_prof.py 5 2>&1 | tail -1
  echo -n "$tag C0: "; timeout -k 10 200 python -u scripts/run_configs.py c0 2>/dev/null | cut -c1-150
done
