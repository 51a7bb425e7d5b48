#!/bin/bash
# C4 soup, per-kernel times of one ray-buffer call at 1 / 4 / 8 / 64 Mi rays (rocprofv3 --kernel-trace --stats): what the
# "fixed cost" of a launch on the big tree consists of.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${1:-c4n}; mkdir -p $OUT
for N in 1 4 8 64; do
  timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n$N -- python3 scripts/c4_prof.py --rays $((N << 20)) --reps 5 > $OUT/n$N.log 2>&1 || { tail -5 $OUT/n$N.log; exit 1; }
  tail -1 $OUT/n$N.log
  python3 - "$OUT/n$N" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "k_bin" in n or "k_trace" in n:
        short = n[n.index("k_"):].split("(")[0]
        print(f"    {short:50s} calls {int(r['Calls']):3d}  avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
done
