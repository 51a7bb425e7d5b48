#!/bin/bash
# Full evidence run for profiles/: default bench line (with CPU baseline), kernel trace, PMC passes, summary.
# usage (on the GPU box): scripts/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r01e}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python3 -u bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -20 gpurun_out/$TAG/bench.err; exit 1; }
tail -c 600 gpurun_out/$TAG/bench.json; echo
export RT_BENCH_QUICK=1      # profiled runs: the frame's kernels only (no CPU legs, no host-pointer transfer figure)
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/kt -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/kt.log 2>&1 || { tail -20 gpurun_out/$TAG/kt.log; exit 1; }
bash scripts/pmc_passes.sh gpurun_out/$TAG/pmc || exit 1
python3 scripts/summarize_prof.py $TAG gpurun_out/$TAG/kt gpurun_out/$TAG/pmc gpurun_out/$TAG/bench.json > gpurun_out/$TAG/summary.md
make -s -C raytracer_amd/csrc resources > profiles/${TAG}_kernel_resources.txt 2>/dev/null
cp profiles/${TAG}_summary.* profiles/${TAG}_kernel_resources.txt profiles/pmc_latest.json gpurun_out/$TAG/ 2>/dev/null
tail -20 gpurun_out/$TAG/summary.md
