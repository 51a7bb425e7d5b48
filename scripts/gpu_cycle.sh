#!/bin/bash
# one GPU round trip: parity tests, then the bench line (no CPU baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/gpu_tests.log | head -20; exit $rc; }
timeout -k 10 300 python -u bench.py --no-cpu-baseline "$@" > gpurun_out/bench_cycle.json 2> gpurun_out/bench_cycle.err || { tail -20 gpurun_out/bench_cycle.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_cycle.json').read().strip().splitlines()[-1])
print("value", d["value"], "Mrays/s  ms/step", d["ms_per_step"])
print("kernels", d["kernels"])
print("roofline", d["roofline"]["frac"], d["roofline"]["work_per_ray"], "hbm-form", d["roofline_hbm"]["frac"])
PY
