#!/bin/bash
# Rehearsal of the multi-rank bench path on ONE GPU: N ranks share device 0, gloo backend, host-staged gather.
# Checks that the gathered RGBA8 frame is identical to the single-rank frame.
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/rehearse_n1.json 2> gpurun_out/rehearse_n1.err || { tail gpurun_out/rehearse_n1.err; exit 1; }
for N in 2 3; do
RT_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/rehearse_n$N.json 2> gpurun_out/rehearse_n$N.err || { tail -20 gpurun_out/rehearse_n$N.err; exit 1; }
done
python - <<'PY'
import json
f=lambda n: json.loads([l for l in open(f'gpurun_out/rehearse_n{n}.json') if l.startswith('{')][-1])
a=f(1)
for n in (2,3):
    b=f(n)
    print(n, 'frame equal:', a['frame_sha256']==b['frame_sha256'], b['value'], b['kernels'])
print(1, a['value'], a['kernels'])
PY
