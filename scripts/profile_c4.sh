#!/bin/bash
# C4 evidence run: kernel trace + PMC passes over scripts/c4_prof.py, summarised into profiles/<tag>_summary.*
set -o pipefail
TAG=${1:-r01f_c4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/kt -- python3 scripts/c4_prof.py > gpurun_out/$TAG/kt.log 2>&1 || { tail -20 gpurun_out/$TAG/kt.log; exit 1; }
tail -1 gpurun_out/$TAG/kt.log
RT_PMC_PROG=scripts/c4_prof.py bash scripts/pmc_passes.sh gpurun_out/$TAG/pmc --reps 1 || exit 1
cp profiles/pmc_latest.json /tmp/pmc_latest.keep
python3 scripts/summarize_prof.py $TAG gpurun_out/$TAG/kt gpurun_out/$TAG/pmc > gpurun_out/$TAG/summary.md
cp /tmp/pmc_latest.keep profiles/pmc_latest.json
cp profiles/${TAG}_summary.* gpurun_out/$TAG/
tail -8 gpurun_out/$TAG/summary.md
