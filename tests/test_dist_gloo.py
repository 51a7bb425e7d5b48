"""CPU tests of the multi-GPU sharding logic (raytracer_amd/dist.py) with world_size-2 gloo groups."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import dist as rd


def test_stripe_plan_covers_every_row_once():
    for h, world, sh in ((2160, 8, 16), (1080, 4, 8), (37, 2, 8), (5, 3, 8), (64, 1, 16)):
        seen = np.zeros(h, dtype=int)
        for r in range(world):
            p = rd.stripe_plan(h, world, r, sh)
            rows = p.local_to_global_rows()
            assert len(rows) == p.padded_rows
            for y in rows:
                if y >= 0:
                    seen[y] += 1
        assert (seen == 1).all(), (h, world, sh)
        idx = rd.gather_index(rd.stripe_plan(h, world, 0, sh), "cpu")
        assert len(set(idx.tolist())) == h


def test_kernel_row_formula_matches_plan():
    """k_primary maps local row yl to y0 + (yl // sh) * sh * step + yl % sh (kernels.hip)."""
    for h, world, sh in ((2160, 8, 16), (37, 2, 8)):
        for r in range(world):
            p = rd.stripe_plan(h, world, r, sh)
            for yl, y in enumerate(p.local_to_global_rows()):
                yk = p.y0 + (yl // sh) * sh * world + yl % sh
                assert (y == yk) if y >= 0 else (yk >= h)


def test_shard_range():
    n, world = 1003, 4
    got = []
    for r in range(world):
        first, cnt, per = rd.shard_range(n, world, r)
        assert per == 251
        got += list(range(first, first + cnt))
    assert got == list(range(n))
    assert rd.shard_range(2, 4, 3) == (2, 0, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, sh, n):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # image: every rank fills its own stripes with f(global pixel), garbage elsewhere
        p = rd.stripe_plan(h, world, rank, sh)
        local = torch.full((p.padded_rows, w, 4), -7.0)
        for yl, y in enumerate(p.local_to_global_rows()):
            if y >= 0:
                local[yl] = (torch.arange(w, dtype=torch.float32)[:, None] + y * w) * torch.tensor([1.0, 2.0, 3.0, 4.0])
        full = rd.gather_rows(local, p)
        want = (torch.arange(h * w, dtype=torch.float32).reshape(h, w, 1)) * torch.tensor([1.0, 2.0, 3.0, 4.0])
        assert torch.equal(full, want)
        for dst in range(world):                       # gather to one rank: the frame on dst, None elsewhere
            one = rd.gather_rows_to(local, p, dst=dst)
            assert (one is None) == (rank != dst)
            if rank == dst:
                assert torch.equal(one, want)
        # the pipelined form: the gather of frame k is in flight while frame k + 1 is written into the other buffer
        fg = rd.FrameGather(p, (w, 4), torch.float32, "cpu", dst=world - 1)
        bufs, prev, frames = [torch.empty_like(local), torch.empty_like(local)], None, []
        for k in range(5):
            bufs[k % 2].copy_(local + 100000.0 * k)
            t = fg.start(bufs[k % 2])
            if prev is not None:
                frames.append(fg.finish(prev))
            prev = t
        frames.append(fg.finish(prev))
        for k, f in enumerate(frames):
            assert (f is None) == (rank != world - 1)
            if f is not None:
                assert torch.equal(f, want + 100000.0 * k), k
        # ray ranges
        first, cnt, per = rd.shard_range(n, world, rank)
        loc = torch.zeros(per, 4)
        loc[:cnt, 0] = torch.arange(first, first + cnt, dtype=torch.float32)
        allr = rd.gather_ranges(loc, n, world)
        assert torch.equal(allr[:, 0], torch.arange(n, dtype=torch.float32))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h,w,sh,n", [(37, 5, 8, 101), (64, 8, 16, 64)])
def test_gather_world2_gloo(h, w, sh, n):
    mp.spawn(_worker, args=(2, _free_port(), h, w, sh, n), nprocs=2, join=True)


def test_init_group_quiet_keeps_stdout_for_the_result_line(tmp_path):
    """dist.init_group_quiet: the backends' start-up banners ("[Gloo] Rank 0 is connected to ...", RCCL's version block) go to
    stderr, so that stdout carries rank 0's one JSON line and nothing else -- what bench.py and scripts/run_c4_ranks.py rely on."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "quiet.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import torch, torch.distributed as dist\n"
        "from raytracer_amd import dist as rd\n"
        "rd.init_group_quiet('gloo')\n"
        "t = torch.ones(1) * (dist.get_rank() + 1)\n"
        "dist.all_reduce(t)\n"
        "if dist.get_rank() == 0:\n"
        "    print('{\"sum\": %d}' % int(t.item()), flush=True)\n"
        "dist.destroy_process_group()\n")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert lines == ['{"sum": 3}'], p.stdout
