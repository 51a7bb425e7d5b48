#!/bin/bash
# Per-kernel times of the C4 workload (10 M-triangle soup, 64 Mi random rays, auto binning): rocprofv3 kernel trace of scripts/c4_prof.py.
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-c4k}
rm -rf gpurun_out/$TAG; mkdir -p gpurun_out/$TAG
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 scripts/c4_prof.py --reps 3 ${@:2} > gpurun_out/$TAG/run.log 2>&1 || { tail -20 gpurun_out/$TAG/run.log; exit 1; }
grep "C4 soup" gpurun_out/$TAG/run.log
python3 - "$TAG" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
for r in csv.DictReader(open(f[0])):
    print("%-90s calls %4s avg %10.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
