"""The N > 1 paths end to end, on the GPU box's one device: `bench.py --gpus 2` run PLAINLY (it starts its own ranks with
torch.distributed.run as a child, before anything touches the GPU) and scripts/run_c4_ranks.py under the launcher.

Both ranks share device 0 and exchange over gloo (RT_BENCH_BACKEND=gloo: results staged through the host), because the
box has one GPU; everything else -- interleaved stripes, decomposition-invariant RNG keys, padded shares, gather to
rank 0, de-interleave, the max-over-ranks clock -- is the code N GPUs run.  The RCCL collectives themselves are covered
by test_rccl_collectives_on_gpu_tensors_world_1 (one rank, "nccl" backend).  SURVEY.md 8(e); the reference has no
counterpart (one VkDevice, src/vulkan/VkRenderer.cpp:1576-1609).

The checks are the ones the driver's scaling run relies on: exit status 0, one JSON line from rank 0, the world size and
exchange it names, and a framebuffer / hit buffer that does not depend on N.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env, timeout=420):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RT_BENCH_QUICK="1", **extra_env)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"{cmd}: rc {p.returncode}\n{p.stderr[-3000:]}"
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), f"stdout carries exactly one line, rank 0's JSON; got {len(lines)}:\n{p.stdout[-2000:]}"
    return json.loads(lines[0])


def test_bench_two_ranks_self_launched_give_the_single_rank_frame():
    base = [sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = _run(base, {})
    two = _run(base + ["--gpus", "2"], {"RT_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and one["ranks"]["world_size"] == 1 and one["ranks"]["exchange"] is None
    assert two["n_gpus"] == 2 and two["ranks"]["world_size"] == 2 and two["ranks"]["backend"] == "gloo"
    assert two["ranks"]["exchange"] == "gather to rank 0"
    assert two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "strong"
    assert one["frame_lit_pixels"] > 1_000_000, "the frame must not be black for its hash to mean anything"
    assert two["frame_sha256"] == one["frame_sha256"], "the gathered frame must not depend on the number of ranks"
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]          # strong scaling: the job is fixed
    assert two["value"] > 0 and two["ms_per_step"] > 0
    # the all-gather form (every rank holds the frame) gives the same picture
    allg = _run(base + ["--gpus", "2"], {"RT_BENCH_BACKEND": "gloo", "RT_BENCH_GATHER": "all"})
    assert allg["ranks"]["exchange"] == "all_gather" and allg["frame_sha256"] == one["frame_sha256"]


def test_bench_exchange_code_on_rccl_with_one_rank():
    """RT_BENCH_FORCE_GROUP=1: bench.py with a one-rank "nccl" (= RCCL) process group on the GPU -- the pipelined exchange exactly as N
    ranks run it (asynchronous gather of frame k into alternating buffers while frame k + 1 is traced, drain, barrier, max over
    ranks) -- with one frame in flight and with three (RT_BENCH_FRAMES=3: what N > 1 ranks run by default -- successive frames
    on three streams with their own path buffers, the exchange on a stream of its own, framebuffers released by events), and the
    blocking and all-gather forms: same frame as the plain run."""
    base = [sys.executable, "bench.py", "--steps", "7", "--warmup", "2", "--no-cpu-baseline"]
    one = _run(base, {})
    assert one["ranks"]["frames_in_flight"] == 1 and one["ranks"]["stage_times_from"] == "timed region"
    for extra, overlapped, exchange, frames in (({}, True, "gather to rank 0", 1), ({"RT_BENCH_FRAMES": "3"}, True, "gather to rank 0", 3),
                                                ({"RT_BENCH_FRAMES": "2"}, True, "gather to rank 0", 2),
                                                ({"RT_BENCH_GATHER_SYNC": "1", "RT_BENCH_FRAMES": "3"}, False, "gather to rank 0", 1),
                                                ({"RT_BENCH_GATHER": "all"}, False, "all_gather", 1)):
        got = _run(base, dict(extra, RT_BENCH_FORCE_GROUP="1"))
        assert got["ranks"]["backend"] == "nccl" and got["ranks"]["world_size"] == 1
        assert got["ranks"]["exchange"] == exchange and got["ranks"]["exchange_overlaps_next_frame"] is overlapped
        assert got["ranks"]["frames_in_flight"] == frames
        assert got["frame_sha256"] == one["frame_sha256"], extra
        assert got["ms_per_step"] < 3 * one["ms_per_step"]
        if frames > 1:                              # stage times of single launches, taken before the timed region
            assert got["ranks"]["stage_times_from"].startswith("3 single frames")
            k, k1 = got["kernels"], one["kernels"]
            assert abs(k["bounce_ms"] - k1["bounce_ms"]) < 0.15 * k1["bounce_ms"], (k, k1)


def test_c4_ray_ranges_over_two_and_three_ranks_give_the_single_rank_hits():
    args = ["--soup", "300000", "--rays", "2000003", "--steps", "2"]          # a ray count no rank count divides
    one = _run([sys.executable, "scripts/run_c4_ranks.py"] + args, {})
    for n, port in ((2, 29731), (3, 29732)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "scripts/run_c4_ranks.py"] + args
        got = _run(cmd, {"RT_BENCH_BACKEND": "gloo"})
        assert got["n_gpus"] == n and got["rays"] == one["rays"]
        assert got["hits_sha256"] == one["hits_sha256"], f"hit buffer gathered from {n} ranks differs from one rank's"
