"""bench.py -- closest-hit throughput of the HIP hot path on the BASELINE.json headline config.

Workload (config.workload): C3 of BASELINE.md -- the "Sponza-class" scene the north_star target
(>= 1 Gray/s closest-hit per MI355X) is quoted on.  The real Sponza mesh is not in the reference and
cannot be fetched, so the scene is the labelled procedural stand-in of SURVEY.md 8(d): 8 x 8
sphereGeometry(45,45) in a closed room = 259,212 triangles.  One step = one 3840x2160 frame at 4 spp,
every stage on device, nothing skipped:
  1. k_primary      jittered primary rays generated on device + closest hit    (33.2 M rays, coherent), fused with
                    segment 0 of the path loop: emission + Cook-Torrance throughput + cosine-weighted bounce ray
                    (frag.glsl:310-350)
  2. k_trace_strip  closest hit of the bounce rays                             (33.2 M rays, incoherent; staged ray queue)
  3. k_shade_resolve  radiance of the second (last) segment + average, gamma -> RGBA8 framebuffer (frag.glsl:357-359)
  4. N > 1 only: gather of the per-rank framebuffer stripes to rank 0 over RCCL/xGMI + de-interleave
value = closest-hit rays of the whole job / wall time (max over ranks).  With N ranks the frame's 16-row
stripes are interleaved over the ranks (BVH replicated): total work is fixed, so scaling is "strong".

Usage: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either launch it with torch.distributed.run (one rank
per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or run it plainly: without WORLD_SIZE in the
environment it starts the N ranks itself, as a torch.distributed.run child process, before anything touches the GPU.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traversal", type=int, default=3, help="2 = BVH2 nodes, 3 = BVH4q nodes")
    return ap.parse_args(argv)


def launcher_command(args, argv, port):
    """The torch.distributed.run command line that starts `args.gpus` ranks of this script on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def needs_self_launch(args, env):
    return args.gpus > 1 and "WORLD_SIZE" not in env


def self_launch(args, argv):
    """N > 1 without a launcher: become the launcher.  Runs before torch.cuda / HIP is touched in this process (the
    ranks are children; nothing here initialises the GPU), prints what rank 0 prints, returns the child's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(launcher_command(args, argv, port), env=env)


if __name__ == "__main__" and needs_self_launch(parse_args(), os.environ):
    sys.exit(self_launch(parse_args(), sys.argv[1:]))

import numpy as np                              # noqa: E402
import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

import raytracer_amd as rt                      # noqa: E402
from raytracer_amd import dist as rd            # noqa: E402
from raytracer_amd import scenes                # noqa: E402

W, H, SPP, SEED, STRIPE = 3840, 2160, 4, 1, 8
# MI355X_MICROARCH.md, chip-level parameters: 157.3 TFLOP/s fp32 vector peak (that figure counts packed
# v_pk_fma_f32; plain wave64 VALU issues one instruction per 4 cycles per SIMD = 78.6 TFLOP/s of scalar fma),
# 8 TB/s HBM3E spec (6.29 TB/s measured achievable).
PEAK_VALU_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
CPU_SAMPLE = 65536            # rays per generation in the CPU-baseline / parity sample
CPU_BVH_SAMPLE = 1 << 24      # rays per generation for the CPU BVH walk (fair CPU row + wide parity gate)
XFER_SAMPLE = 1 << 24         # rays of the transfer-inclusive figure (512 MB in, 256 MB out)


def flops_bytes_per_ray(c, generated):
    """flops/ray = 24 B + 53 T + 45 G; bytes/ray = 64 N + 48 T + IO."""
    rays = max(c["rays"], 1)
    B, N, T = c["box_tests"] / rays, c["nodes"] / rays, c["tri_tests"] / rays
    flops = 24.0 * B + 53.0 * T + (45.0 if generated else 0.0)
    nbytes = 64.0 * N + 48.0 * T + (16.0 if generated else 48.0)
    return flops, nbytes, {"B": round(B, 2), "N": round(N, 2), "T": round(T, 2)}


def main():
    args = parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:                       # launched by torch.distributed.run: the launcher's world size rules
        args.gpus = world
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")             # "gloo": rehearsal of N ranks on ONE GPU
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_GROUP=1: a process group even for ONE rank, so that the exchange code N ranks run -- asynchronous RCCL gather
    # into alternating buffers, drain, barrier, max over ranks -- executes on a one-GPU box (tests/test_gpu_ranks.py)
    grouped = world > 1 or os.environ.get("RT_BENCH_FORCE_GROUP") == "1"
    if grouped and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if grouped:
        rd.init_group_quiet(backend, dev)        # RCCL / gloo banners go to stderr: stdout carries rank 0's JSON line only

    # ---- scene setup: flatten + upload to this rank's HBM (BVH replicated per GPU) -------------------
    tris, sph, lights, cam = scenes.sponza_class()
    r = rt.Renderer((local_rank,))
    r.set_option(rt.OPT_TRAVERSAL, args.traversal)
    for kv in filter(None, os.environ.get("RT_BENCH_OPTS", "").split(",")):      # tuning sweeps only: key=value pairs of rt_set_option
        r.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    stats = r.create_data(tris, sph, lights)
    r.update_camera(cam)
    # which primary kernel the library picks for this tree (include/raytracer_amd.h, RT_OPT_PRIMARY)
    forced = {kv.split("=")[0]: kv.split("=")[1] for kv in filter(None, os.environ.get("RT_BENCH_OPTS", "").split(","))}.get("10", "0")
    primary_kernel = ({"1": "k_primary", "2": "ray generation + k_trace_strip", "3": "k_primary_packet"}.get(forced) or
                      ("k_primary_packet" if args.traversal == 3 else "ray generation + k_trace_strip" if stats["sah_cost"] > 250 else "k_primary"))

    # RT_BENCH_SHARE_OF=M (measurement aid, one rank): this GPU runs rank 0's share of a world of M ranks -- its stripes, its
    # launches, the exchange of its share through a one-rank group -- with this file's own loop; the line then describes that
    # share (`emulated_share_of`), not the job.  Source of the predicted scaling table of DESIGN.md 6.
    share_of = int(os.environ.get("RT_BENCH_SHARE_OF", "0")) if world == 1 else 0
    stripe_world = share_of or world
    plan = rd.stripe_plan(H, stripe_world, rank, STRIPE)
    xplan = plan if not share_of else rd.stripe_plan(plan.padded_rows, 1, 0, STRIPE)      # what the exchange moves
    n_local = plan.padded_rows * W * SPP                 # local ray slots (padding rows are dead rays)
    n_px = plan.padded_rows * W
    rays_per_step = W * H * SPP * 2                      # whole job: primary + bounce closest-hit rays
    if share_of:
        rays_per_step = plan.local_stripes * STRIPE * W * SPP * 2
    f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
    gather_all = os.environ.get("RT_BENCH_GATHER", "root") == "all"          # "all": every rank gets the frame (all-gather)
    # RCCL, gather to rank 0: the exchange of frame k is asynchronous (RCCL's own stream) and overlaps the kernels of frame
    # k + 1; frame k is complete on rank 0 -- waited for and de-interleaved -- by the end of step k + 1, the last one by the
    # drain inside the timed region.  RT_BENCH_GATHER_SYNC=1 keeps the blocking form for A/B runs.
    pipelined = grouped and backend == "nccl" and not gather_all and os.environ.get("RT_BENCH_GATHER_SYNC") != "1"
    # Frames in flight (N >= 8 with the pipelined exchange): successive frames alternate between two streams with their own path
    # buffers, so that the kernels of frame k + 1 fill the end of frame k's launches -- the slowest rays of a launch finish with
    # the chip half empty, a fixed ~0.2 ms per frame that is 3 % of a whole frame but a quarter of one rank's share at N = 8.
    # Measured with this loop on one GPU (RT_BENCH_SHARE_OF, scripts/share_bench.sh, profiles/r05l_share_bench.txt): the share of
    # one rank of 8 takes 0.81 ms per frame with one frame in flight and 0.76 with two; a quarter of the frame gains 1 %, a half
    # and the whole frame LOSE 1-3 % with two (the launches are long against their tails, and two persistent kernels side by
    # side take each other's cache), so fewer than 8 ranks keep one.  Two, not three: HIP maps a process's streams onto
    # 4 hardware queues and a third stream shares its queue with the first (kernel trace: scripts/frames_trace.sh).
    # K timed steps still deliver K complete frames on rank 0.
    F = max(1, int(os.environ.get("RT_BENCH_FRAMES", "2" if (pipelined and stripe_world >= 8) else "1"))) if pipelined else 1
    rays, phits = f32(n_local * 8), f32(n_local * 4)                 # unfused primary pass (instrumentation, parity sample)
    path = [dict(brays=f32(n_local * 8), bhits=f32(n_local * 4), thr=f32(n_local * 4), rad=f32(n_local * 4)) for _ in range(F)]
    brays, bhits = path[0]["brays"], path[0]["bhits"]
    # RGBA8 framebuffer stripes of this rank; F + 1 of them for N > 1, so that the gather of frame k can run while the
    # frames after it are traced into the others
    fbs = [torch.empty(n_px, dtype=torch.int32, device=dev) for _ in range(F + 1 if grouped else 1)]
    gidx = rd.gather_index(xplan, dev) if grouped else None
    state = {"k": 0, "ticket": None, "frame": None, "ticket_fb": None}
    cur_fb = lambda: fbs[state["k"] % len(fbs)]
    cur = lambda: path[state["k"] % F]
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] if F == 1 else [torch.cuda.Stream(device=dev) for _ in range(F)]
    xs = main_stream if F == 1 else torch.cuda.Stream(device=dev)     # the exchange: gathers and de-interleaves, in frame order
    cur_stream = lambda: streams[state["k"] % F]
    fb_free = [None] * len(fbs)                                       # event: the gather that read this framebuffer is complete
    row_len = W * SPP
    key0 = plan.y0 * row_len                                         # RNG key = path index in the full frame

    def stage_primary_shade():
        # primary rays, their closest hits and segment 0 of the path loop in one launch: the primary rays and
        # hits stay in registers, throughput / radiance / bounce rays are written once
        b = cur()
        r.primary_shade_stripes_dev(W, H, plan.y0, plan.padded_rows, STRIPE, stripe_world, b["thr"], b["rad"], b["brays"], spp=SPP, seed=SEED,
                                    jitter=True, shade_seed=SEED + 1, stream=cur_stream())

    def unfused_primary():
        # same rays and hits through the separate entry point: instrumented pass and parity sample only (untimed)
        r.trace_primary_stripes_dev(W, H, plan.y0, plan.padded_rows, STRIPE, stripe_world, phits, rays, spp=SPP, seed=SEED,
                                    jitter=True, stream=main_stream)

    def stage_bounce_trace():
        b = cur()
        r.trace_dev(b["brays"], n_local, b["bhits"], stream=cur_stream())

    def stage_shade2_resolve():
        # radiance of the second (last) segment + resolve in one launch: the final radiance stays in registers
        b = cur()
        free = fb_free[state["k"] % len(fbs)]
        if free is not None:                      # F > 1: the gather of the frame that used this framebuffer last must be through
            cur_stream().wait_event(free)
        r.shade_resolve_dev(b["brays"], b["bhits"], n_local, SPP, b["thr"], b["rad"], None, cur_fb(), stream=cur_stream())

    fg = rd.FrameGather(xplan, (W,), torch.int32, dev, 0, gidx) if pipelined else None

    def stage_gather():
        # per-rank RGBA8 stripes -> the frame on rank 0 (what shows it): each rank sends its 4 MB once, over its own
        # xGMI link to rank 0 (RCCL send/recv), and rank 0 de-interleaves the stripes with one index_select
        if not grouped:
            state["frame"] = cur_fb()
            return
        mine = cur_fb().view(plan.padded_rows, W)
        if pipelined:
            if F > 1:                             # the exchange stream takes over behind this frame's kernels
                done = torch.cuda.Event()
                done.record(cur_stream())
                xs.wait_event(done)
            with torch.cuda.stream(xs):
                ticket = fg.start(mine)
                if state["ticket"] is not None:
                    state["frame"] = fg.finish(state["ticket"])
                    if F > 1:
                        ev = torch.cuda.Event(); ev.record(xs); fb_free[state["ticket_fb"]] = ev
            state["ticket"], state["ticket_fb"] = ticket, state["k"] % len(fbs)
            return
        if backend != "nccl":                                                    # rehearsal: stage through the host
            mine, idx = mine.cpu(), gidx.cpu()
        else:
            idx = gidx
        state["frame"] = rd.gather_rows(mine, xplan, idx) if gather_all else rd.gather_rows_to(mine, xplan, 0, idx)

    def drain():
        if state["ticket"] is not None:
            with torch.cuda.stream(xs):
                state["frame"] = fg.finish(state["ticket"])
                if F > 1:
                    ev = torch.cuda.Event(); ev.record(xs); fb_free[state["ticket_fb"]] = ev
            state["ticket"] = None

    stages = [stage_primary_shade, stage_bounce_trace, stage_shade2_resolve, stage_gather]

    def step():
        for st in stages:
            st()
        state["k"] += 1

    # ---- instrumented pass: B, N, T per ray (same kernels with counters on) --------------------------
    r.set_option(rt.OPT_COUNTERS, 1)
    r.counters()
    unfused_primary(); torch.cuda.synchronize()
    c_primary = r.counters()
    stage_primary_shade(); stage_bounce_trace(); torch.cuda.synchronize()
    c_bounce = r.counters()
    r.set_option(rt.OPT_COUNTERS, 0)

    # ---- warmup, then exactly K timed steps ------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    drain()
    ne = len(stages) + 1
    probe = None
    if F > 1 or stripe_world > 1:
        # N > 1: per-stage times of single launches from three frames run one at a time BEFORE the timed region, and no event
        # records in it.  With F > 1 the stages of neighbouring frames overlap there (events around one stage would time its
        # neighbours too), and on a share of 0.8 ms the five records per step are themselves worth 4 % (share of one rank of 8,
        # two frames in flight: 0.798 -> 0.762 ms; profiles/r05l_share_bench.txt).  One rank keeps its events in the timed region.
        probe = [[torch.cuda.Event(enable_timing=True) for _ in range(ne)] for _ in range(3)]
        for pe in probe:
            torch.cuda.synchronize()
            pe[0].record(cur_stream())
            for j, st in enumerate(stages):
                st_stream = cur_stream()
                st()
                pe[j + 1].record(st_stream)
            state["k"] += 1
            drain()
        torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(ne)] for _ in range(args.steps)]
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if probe is not None:                            # N > 1: stage times were taken above; no event records between the launches
            step()
            continue
        st_stream = cur_stream()
        ev[k][0].record(st_stream)
        for j, st in enumerate(stages):
            st()
            ev[k][j + 1].record(st_stream)
        state["k"] += 1
    t_enqueued = time.perf_counter() - t0                # host time to queue the K steps (the GPU runs behind it)
    drain()                                              # the last frame's gather + de-interleave belong to the K steps
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    tev = probe if probe is not None else ev
    ms = [float(np.mean([e[j].elapsed_time(e[j + 1]) for e in tev])) for j in range(len(stages))]
    ms_primary, ms_btrace, ms_shade2, ms_gather = ms

    step()                                               # one more (untimed) frame for the checksum below
    drain()
    frame = state["frame"]
    unfused_primary()                                    # primary rays / hits in HBM for the parity sample
    torch.cuda.synchronize()
    if rank != 0:
        if grouped:
            dist.destroy_process_group()
        return
    frame_np = frame.reshape(-1)[: W * (xplan.h if share_of else H)].cpu().numpy()
    frame_sha = hashlib.sha256(frame_np.tobytes()).hexdigest()
    lit_pixels = int((frame_np.view("uint32") & 0xFFFFFF != 0).sum())       # the hash says something only if the frame is not black

    value = rays_per_step * args.steps / elapsed / 1e6
    n_real_local = c_primary["rays"]                     # live rays this rank traces per generation
    fl_p, by_p, w_p = flops_bytes_per_ray(c_primary, True)
    fl_b, by_b, w_b = flops_bytes_per_ray(c_bounce, False)
    live_b = c_bounce["rays"]
    # dominant kernel = the incoherent bounce closest-hit launch (k_trace_strip)
    ach_tflops = fl_b * live_b / (ms_btrace * 1e-3) / 1e12
    ach_gbs = by_b * live_b / (ms_btrace * 1e-3) / 1e9
    # Hardware counters cannot be read inside this process: `traffic` and the measured_* fields are PASTED from the last
    # rocprofv3 --pmc passes over this same command (scripts/profile_round.sh -> profiles/pmc_latest.json).  The file names
    # the kernels.hip it was measured on; if that is not the source this tree holds, the fields are null, not stale.
    pmc, pmc_src = {}, {"pmc_source": None, "pmc_tag": None, "pmc_matches_kernel_sources": None}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path) and world == 1:
        try:
            pmc = json.load(open(pmc_path))
            ksha = hashlib.sha256(b"".join(open(os.path.join(ROOT, "raytracer_amd", "csrc", f), "rb").read() for f in ("kernels.hip", "kernels.hpp", "Makefile"))).hexdigest()
            ok = pmc.get("kernel_sources_sha256") == ksha
            pmc_src = {"pmc_source": "profiles/pmc_latest.json (rocprofv3 --pmc passes of this command, separate run)",
                       "pmc_tag": pmc.get("tag"), "pmc_matches_kernel_sources": bool(ok)}
            if not ok:
                pmc = {}
        except Exception:
            pmc = {}
    kname = "k_trace_strip<%s> (bounce closest-hit)" % ("bvh4q" if args.traversal == 3 else "bvh2")
    # Two ceilings, both reported: fp32 VALU and HBM.  On this workload the structure (17 MB) is
    # L2/MALL resident: measured HBM traffic (`traffic`, rocprofv3 PMC, profiles/) is a small fraction of the
    # algorithmic bytes, and the kernel is bound by VALU issue, so `roofline` is the VALU one; the byte-based
    # figure is kept beside it and may exceed the HBM peak precisely because those bytes are cache hits.
    roofline = {"bound": "valu", "kernel": kname, "achieved": round(ach_tflops, 3),
                "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s", "frac": round(ach_tflops / PEAK_VALU_TFLOPS, 4),
                "traffic": pmc.get("k_trace_hbm_bytes_per_launch"), "flops_per_ray": round(fl_b, 1), "work_per_ray": w_b,
                "launch_ms": round(ms_btrace, 4), "rays_per_launch": int(live_b),
                "measured_valu_busy_pct": pmc.get("k_trace_valu_busy_pct"),
                "measured_valu_inst_per_simd_cycle": pmc.get("k_trace_valu_inst_per_simd_cycle"),
                "measured_lane_utilisation_pct": pmc.get("k_trace_lane_utilisation_pct"), **pmc_src,
                "note": "SURVEY 8d flop count (24 per box test, 53 per triangle test) over the HIP-event launch time; "
                        "never MFMA.  Most issued VALU instructions are converts/compares/selects, so the issue-slot "
                        "utilisation is measured_valu_busy_pct x measured_lane_utilisation_pct (rocprofv3, profiles/)"}
    roofline_hbm = {"bound": "hbm", "kernel": kname, "achieved": round(ach_gbs, 1), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(ach_gbs / PEAK_HBM_GBS, 4), "traffic": pmc.get("k_trace_hbm_bytes_per_launch"),
                    "algorithmic_bytes_per_launch": round(by_b * live_b), "bytes_per_ray": round(by_b, 1),
                    "note": "algorithmic bytes (64 B per node visit + 48 B per triangle test + 48 B ray/hit IO) are an "
                            "upper bound on HBM traffic; here they are served by L2 (hit rate in profiles/), so this "
                            "fraction is not a utilisation and can exceed 1"}

    out = {
        "metric": "closest-hit Mrays/s", "value": round(value, 1), "unit": "Mrays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "host_enqueue_ms_per_step": round(t_enqueued / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C3 sponza-class stand-in (8x8 sphereGeometry(45,45) in a room, 259212 tris), "
                               "3840x2160 x 4 spp jittered primary + 1 cosine-weighted bounce, closest-hit + shading "
                               "to an RGBA8 framebuffer",
                   "rays_per_step": rays_per_step, "parallelism": f"stripes{STRIPE}x{stripe_world}",
                   **({"emulated_share_of": share_of} if share_of else {}),
                   "bvh": {"layout": "bvh4q" if args.traversal == 3 else "bvh2", "nodes": stats["n_nodes"],
                           "depth": stats["max_depth"], "node_bytes": stats["node_bytes"],
                           "tri_bytes": stats["tri_bytes"], "build_ms": round(stats["build_ms"], 1)}},
        "frame_sha256": frame_sha,       # RGBA8 framebuffer: identical for every N (decomposition-invariant RNG keys)
        "frame_lit_pixels": lit_pixels,
        "ranks": {"world_size": dist.get_world_size() if grouped else 1,
                  "backend": (dist.get_backend() if grouped else None),          # "nccl" = RCCL over xGMI
                  "exchange": None if not grouped else ("all_gather" if gather_all else "gather to rank 0"),
                  "exchange_overlaps_next_frame": bool(pipelined),
                  # F > 1: the kernels of successive frames overlap on F streams.  N > 1: the per-stage times below come from three
                  # frames run one at a time before the timed region
                  "frames_in_flight": F, "stage_times_from": "timed region" if probe is None else "3 single frames before the timed region"},
        # dynamic LDS of the traversal kernels (a launch argument: profilers show 0): [stack entry][256 threads] ints, and for
        # the ray-buffer kernel the staged ray queues behind them (4 waves x 32 rays x 64 B; rt_api.cpp trace_rays)
        # (the packet form of the tile kernel keeps its stack in a register: no LDS)
        "launch": {"dynamic_lds_bytes_per_block": {"k_primary_packet": 0,
                                                   "k_primary": stats["stack_entries"] * 1024,
                                                   "k_trace_strip": min(stats["stack_entries"], 12) * 1024 + 8192},
                   "primary_kernel": primary_kernel,
                   "stack_entries": stats["stack_entries"], "lds_stack_entries": 12, "ray_queue_bytes_per_block": 8192},
        "roofline": roofline, "roofline_hbm": roofline_hbm,
        "kernels": {"primary_Mrays_s": round(n_real_local / ms_primary / 1e3, 1), "primary_shade_ms": round(ms_primary, 4),
                    "primary_work_per_ray": w_p, "primary_tflops": round(fl_p * n_real_local / (ms_primary * 1e-3) / 1e12, 3),
                    "bounce_Mrays_s": round(live_b / ms_btrace / 1e3, 1), "bounce_ms": round(ms_btrace, 4),
                    "shade2_resolve_ms": round(ms_shade2, 4), "gather_ms": round(ms_gather, 4)},
    }

    # ---- CPU baseline (oracle = reference semantics, brute force) + parity gate on the same sample --------
    quick = os.environ.get("RT_BENCH_QUICK") == "1"   # kernel A/B runs: skip the CPU legs and the transfer figure
    if not args.no_cpu_baseline and world == 1 and not quick:       # the CPU legs run on rank 0 at N = 1 only
        import oracle
        torch.cuda.synchronize()
        live = plan.local_stripes * STRIPE * W * SPP if world > 1 else W * H * SPP
        live = min(live, n_local)
        idx = np.sort(np.random.default_rng(0).choice(live, CPU_SAMPLE, replace=False))
        tidx = torch.from_numpy(idx).to(dev)
        take = lambda buf, k: buf.view(-1, k).index_select(0, tidx).cpu().numpy()
        s_rays = take(rays, 8).view(rt.RAY).ravel()
        s_brays = take(brays, 8).view(rt.RAY).ravel()
        s_hits = take(phits, 4).view(rt.HIT).ravel()
        s_bhits = take(bhits, 4).view(rt.HIT).ravel()
        sc = oracle.Scene(tris, sph)
        tc = time.perf_counter()
        o_hits = oracle.trace(sc, s_rays)
        o_bhits = oracle.trace(sc, s_brays)
        cpu_s = time.perf_counter() - tc
        mism = int((o_hits["prim"] != s_hits["prim"]).sum() + (o_bhits["prim"] != s_bhits["prim"]).sum())
        exact = (o_hits.tobytes() == s_hits.tobytes()) and (o_bhits.tobytes() == s_bhits.tobytes())
        out["cpu_baseline"] = {"value": round(2 * CPU_SAMPLE / cpu_s / 1e6, 4), "unit": "Mrays/s",
                               "cores": oracle.num_threads(), "kind": "port",
                               "sample": f"{CPU_SAMPLE} primary + {CPU_SAMPLE} bounce rays of the same frame (seeded "
                                         f"subset), brute force over all 259212 triangles, {cpu_s:.1f} s"}
        out["parity"] = {"sample_rays": 2 * CPU_SAMPLE, "prim_id_mismatches": mism, "hit_records_bit_exact": bool(exact)}
        # SURVEY 8(d) second CPU row, "build CPU BVH, not reference": the oracle walks the BVH2 that the library's host
        # builder flattens (handed over as data).  Fair CPU figure, and a wider parity gate (its walk is tested equal
        # to the brute-force loop in tests/test_bvh_host.py).
        _, nodes2, order2 = rt.bvh_build_host(tris, layout=rt.TRAVERSAL_BVH2)
        m = min(CPU_BVH_SAMPLE, live)
        w_rays = rays.view(-1, 8)[:m].cpu().numpy().view(rt.RAY).ravel()
        w_brays = brays.view(-1, 8)[:m].cpu().numpy().view(rt.RAY).ravel()
        tc = time.perf_counter()
        w_hits = oracle.trace_bvh2(sc, nodes2, order2, w_rays)
        w_bhits = oracle.trace_bvh2(sc, nodes2, order2, w_brays)
        walk_s = time.perf_counter() - tc
        g_hits = phits.view(-1, 4)[:m].cpu().numpy().view(rt.HIT).ravel()
        g_bhits = bhits.view(-1, 4)[:m].cpu().numpy().view(rt.HIT).ravel()
        out["cpu_bvh_baseline"] = {"value": round(2 * m / walk_s / 1e6, 3), "unit": "Mrays/s", "cores": oracle.num_threads(),
                                   "kind": "build CPU BVH, not reference",
                                   "sample": f"first {m} primary + {m} bounce rays of the frame, BVH2 walk, {walk_s:.1f} s"}
        out["parity"].update({"bvh_walk_sample_rays": 2 * m,
                              "bvh_walk_prim_id_mismatches": int((w_hits["prim"] != g_hits["prim"]).sum() + (w_bhits["prim"] != g_bhits["prim"]).sum()),
                              "bvh_walk_bit_exact": bool(w_hits.tobytes() == g_hits.tobytes() and w_bhits.tobytes() == g_bhits.tobytes())})
    # ---- SURVEY 8(d) second figure: closest hit INCLUDING the transfers (host rays in, host hits out) ----------------
    # What the reference's frame loop would call with host arrays (rt_trace): 2 Mi-ray pieces, H2D | kernel | D2H
    # overlapped on four streams (one per copy direction, two for the kernels).  Never `value`.  The PCIe bound is 32 B in + 16 B out per ray over the measured link rates.
    if world == 1 and not quick:
        m = min(n_local, XFER_SAMPLE)
        h_rays = brays.view(-1, 8)[:m].cpu().numpy().view(rt.RAY).ravel()
        want = bhits.view(-1, 4)[:m].cpu().numpy().view(rt.HIT).ravel()
        p_rays, p_hits = r.host_alloc(m, rt.RAY), r.host_alloc(m, rt.HIT)
        p_rays[:] = h_rays
        pin_t = torch.from_numpy(p_rays.view(np.float32))              # pinned by rt_host_alloc
        d_t = torch.empty(m * 8, dtype=torch.float32, device=dev)

        def gbs(fn, nbytes, reps=3):
            fn(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return nbytes * reps / (time.perf_counter() - t) / 1e9
        h2d = gbs(lambda: d_t.copy_(pin_t, non_blocking=True), m * 32)
        d2h = gbs(lambda: pin_t.copy_(d_t, non_blocking=True), m * 32)
        p_rays[:] = h_rays
        res = {}
        for name, src, dst in (("pinned", p_rays, p_hits), ("pageable", h_rays, np.empty(m, dtype=rt.HIT))):
            r.trace(src, out=dst)                                      # warm: staging buffers, scratch
            t = time.perf_counter()
            r.trace(src, out=dst)
            dt = time.perf_counter() - t
            res[name + "_Mrays_s"] = round(m / dt / 1e6, 1)
            res[name + "_bit_exact"] = bool(dst.tobytes() == want.tobytes())
        bound = 1.0 / (32.0 / (h2d * 1e9) + 0.0) / 1e6                 # full duplex: the 32 B/ray inbound leg is the longer one
        res.update({"rays": int(m), "h2d_GBs": round(h2d, 1), "d2h_GBs": round(d2h, 1), "pcie_bound_Mrays_s": round(bound, 1),
                    "pinned_frac_of_pcie_bound": round(res["pinned_Mrays_s"] / bound, 3),
                    "note": "rt_trace on the bounce rays of the frame: host rays in (32 B), host hits out (16 B); pinned = blocks of "
                            "rt_host_alloc, pageable = ordinary memory staged through pinned buffers by host threads"})
        out["with_transfers"] = res
        r.host_free(p_rays); r.host_free(p_hits)
    print(json.dumps(out), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
