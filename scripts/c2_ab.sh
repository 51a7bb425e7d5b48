#!/bin/bash
# C2 (1920x1080 x 1 spp primary + shadow rays) per library variant on one box.  usage: scripts/c2_ab.sh <tag> ...  ("base" = the in-tree library)
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=raytracer_amd/csrc/libraytracer_amd.so
  [ "$tag" != base ] && lib=raytracer_amd/csrc/libraytracer_amd_$tag.so
  for rep in 1 2; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python -u scripts/run_configs.py c2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['bvh4q']; print('$tag rep$rep', {k:b[k] for k in b if k.endswith('_ms')}, b['work_primary'])" || echo "$tag failed"
  done
done
