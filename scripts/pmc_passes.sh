#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload, one pass per run (gfx950 slot limits:
# 8 SQ, 4 TCC, 4 TCP, 2 TA, 2 GRBM per pass; FETCH_SIZE = 3 TCC, WRITE_SIZE = 2 TCC).
# usage: [RT_PMC_PROG=scripts/c4_prof.py] scripts/pmc_passes.sh <outdir> [program args...]
set -o pipefail
OUT=${1:-gpurun_out/pmc}; shift
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline}
PROG=${RT_PMC_PROG:-bench.py}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
pass() {
  local name=$1; shift
  echo "== pass $name: $*"
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $PROG $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT &&
pass sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS &&
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
echo "pmc exit=$?"
