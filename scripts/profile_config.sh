#!/bin/bash
# Counter evidence for one config: kernel trace + the five PMC passes over a workload script, summarised into
# profiles/<tag>_summary.{md,json} (copied to gpurun_out/<tag>/ for the way back).  profiles/pmc_latest.json (the bench line's
# source) is left as it is.
# usage (on the GPU box): scripts/profile_config.sh <tag> <program.py> <program args...>
set -o pipefail
TAG=$1; PROG=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/kt -- python3 $PROG "$@" > gpurun_out/$TAG/kt.log 2>&1 || { tail -20 gpurun_out/$TAG/kt.log; exit 1; }
tail -1 gpurun_out/$TAG/kt.log
RT_PMC_PROG=$PROG bash scripts/pmc_passes.sh gpurun_out/$TAG/pmc "$@" || exit 1
cp profiles/pmc_latest.json /tmp/pmc_latest.keep
python3 scripts/summarize_prof.py $TAG gpurun_out/$TAG/kt gpurun_out/$TAG/pmc > gpurun_out/$TAG/summary.md
cp /tmp/pmc_latest.keep profiles/pmc_latest.json
cp profiles/${TAG}_summary.* gpurun_out/$TAG/
tail -12 gpurun_out/$TAG/summary.md
