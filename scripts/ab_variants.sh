#!/bin/bash
# A/B of kernel builds on ONE box: runs the quick bench (no CPU legs) once per library variant and prints the stage times.
# usage: scripts/ab_variants.sh <tag> [<tag> ...]   ("base" = libraytracer_amd.so, else libraytracer_amd_<tag>.so)
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=raytracer_amd/csrc/libraytracer_amd.so
  [ "$tag" != base ] && lib=raytracer_amd/csrc/libraytracer_amd_$tag.so
  [ -f "$lib" ] || { echo "$tag: $lib missing"; continue; }
  for rep in 1 2; do
    RT_AMD_LIB=$PWD/$lib RT_BENCH_QUICK=1 timeout -k 10 120 python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$tag rep$rep: step %.3f ms  primary %.3f  bounce %.3f  shade2 %.3f  N=%.2f sha %s' % (d['ms_per_step'], k['primary_shade_ms'], k['bounce_ms'], k['shade2_resolve_ms'], d['roofline']['work_per_ray']['N'], d['frame_sha256'][:12]))" || echo "$tag rep$rep failed"
  done
done
