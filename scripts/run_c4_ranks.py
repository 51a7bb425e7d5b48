"""C4 over N ranks: BVH replicated, the ray buffer cut into N contiguous ranges, hits all-gathered.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         scripts/run_c4_ranks.py [--soup 10000000 --rays 67108864 --steps 3]
RT_BENCH_BACKEND=gloo rehearses N ranks on ONE GPU (hits staged through the host).  Rank 0 prints one JSON line:
whole-job Mrays/s (all ranks' rays / max time, gather included) and a checksum of the gathered hit buffer, which must
not depend on N.
"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import raytracer_amd as rt
from raytracer_amd import dist as rd
from raytracer_amd import scenes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--soup", type=int, default=10_000_000)
    ap.add_argument("--rays", type=int, default=64 << 20)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        rd.init_group_quiet(backend, dev)
    tris = scenes.soup(a.soup)
    first, count, per = rd.shard_range(a.rays, world, rank)
    rays_np = np.zeros(per, dtype=rt.RAY)                                  # padded to `per`: dead rays (tmax 0) past the end
    if count:
        rays_np[:count] = scenes.random_rays(count, first=first)
    rays = torch.from_numpy(rays_np.view(np.float32).reshape(-1)).to(dev)
    hits = torch.empty(per * 4, dtype=torch.float32, device=dev)
    r = rt.Renderer((local_rank,))
    st = r.create_data(tris)
    stream = torch.cuda.current_stream()

    def step():
        r.trace_dev(rays, per, hits, stream=stream)
        h = hits.view(per, 4)
        if world > 1 and backend != "nccl":
            return rd.gather_ranges(h.cpu(), a.rays, world)
        return rd.gather_ranges(h, a.rays, world)

    step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        sha = hashlib.sha256(out.contiguous().cpu().numpy().tobytes()).hexdigest()
        print(json.dumps({"config": "C4", "n_gpus": world, "tris": a.soup, "rays": a.rays, "steps": a.steps,
                          "ms_per_step": round(el / a.steps * 1e3, 3), "Mrays_s": round(a.rays * a.steps / el / 1e6, 1),
                          "bvh_build_ms": round(st["build_ms"], 1), "hits_sha256": sha}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
