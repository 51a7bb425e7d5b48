#!/bin/bash
# kernel trace of bench.py with a one-rank RCCL group: share of M ranks (default 8), F frames in flight; prints the timeline of a few
# frames in the middle of the timed region.  usage: scripts/frames_trace.sh [M] [F ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
M=${1:-8}; shift; FS=${@:-1 3}
export RT_BENCH_FORCE_GROUP=1 RT_BENCH_QUICK=1 RT_BENCH_SHARE_OF=$M
for F in $FS; do
  export RT_BENCH_FRAMES=$F
  rm -rf gpurun_out/frames_f$F
  timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/frames_f$F -- python3 bench.py --steps 30 --warmup 4 --no-cpu-baseline > gpurun_out/frames_f$F.log 2>&1 || { tail -20 gpurun_out/frames_f$F.log; exit 1; }
  grep '^{' gpurun_out/frames_f$F.log | cut -c1-160
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/frames_f$F/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_trace_strip" in r["Kernel_Name"]]
a = idx[len(idx) // 2]; t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a - 2:a + 34]:
    n = r["Kernel_Name"]; n = n[n.find("k_"):n.find("<")] if "::k_" in n else n[:34]
    print("  %-34s q%-2s start %8.1f  end %8.1f  dur %7.1f us" % (n, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
done
