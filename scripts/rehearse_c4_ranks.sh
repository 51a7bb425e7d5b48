#!/bin/bash
# C4 ray-range sharding rehearsed on ONE GPU: 1, 2 and 3 gloo ranks on a reduced soup; the gathered hit buffers must match.
cd "$GRAFT_REPO_ROOT" || exit 1
ARGS="--soup 400000 --rays 3000001 --steps 2"
timeout -k 10 300 python -u scripts/run_c4_ranks.py $ARGS > gpurun_out/c4r_n1.json 2> gpurun_out/c4r_n1.err || { tail gpurun_out/c4r_n1.err; exit 1; }
for N in 2 3; do
RT_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2961$N scripts/run_c4_ranks.py $ARGS > gpurun_out/c4r_n$N.json 2> gpurun_out/c4r_n$N.err || { tail -20 gpurun_out/c4r_n$N.err; exit 1; }
done
python - <<'PY'
import json
f=lambda n: json.loads([l for l in open(f'gpurun_out/c4r_n{n}.json') if l.startswith('{')][-1])
a=f(1); print(1, a)
for n in (2,3):
    b=f(n); print(n, 'hits equal:', a['hits_sha256']==b['hits_sha256'], b['Mrays_s'])
PY
