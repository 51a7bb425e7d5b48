#!/bin/bash
cd "$GRAFT_REPO_ROOT"
scripts/ab_variants.sh base pnt base pnt
for l in "" _pnt; do echo -n "C5 lib$l: "; RT_AMD_LIB=$PWD/raytracer_amd/csrc/libraytracer_amd$l.so timeout -k 10 200 python -u scripts/c5_stages.py 2>&1 | tail -1 | cut -c1-200; done
