"""Multi-GPU sharding of the hot path: one process per GPU, scene replicated, work split, one gather.

The reference is single-device (one VkDevice, src/vulkan/VkRenderer.cpp:1576-1609); everything here is
build-original.  Rays and pixels are independent (shaders/frag.glsl:303-360 has no
cross-pixel communication), so the only exchange is the gather of per-rank result slices:

  * image configs: the frame is cut into horizontal stripes of `stripe_h` rows (a multiple of the 8x8
    wave tile); rank r owns stripes r, r+N, r+2N, ... (interleaved, because cost is spatially uneven);
    the per-rank buffers are gathered to one rank (gather_rows_to) or all-gathered (gather_rows) -- RCCL
    over xGMI when the backend is "nccl" -- and de-interleaved into image order.
  * ray-buffer configs (C4): contiguous ranges of ceil(n/N) rays per rank, all-gathered the same way.

No reduction is involved, so no all-reduce: each element has exactly one owner.
torch.distributed is plumbing here (process group + collective); the math is index bookkeeping.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class StripePlan:
    h: int            # image rows
    world: int
    rank: int
    stripe_h: int     # rows per stripe
    n_stripes: int    # stripes in the whole image (last one may be short)
    max_local: int    # stripes owned by rank 0 (= the padded per-rank count)

    @property
    def y0(self):
        return self.rank * self.stripe_h

    @property
    def local_stripes(self):
        return len(range(self.rank, self.n_stripes, self.world))

    @property
    def padded_rows(self):
        """Rows of the per-rank buffer: identical on every rank so that all_gather needs no sizes."""
        return self.max_local * self.stripe_h

    def local_to_global_rows(self, rank=None):
        """Image row of every local row of `rank` (rows past the image bottom map to -1)."""
        rank = self.rank if rank is None else rank
        rows = []
        for k in range(self.max_local):
            s = rank + k * self.world
            for yy in range(self.stripe_h):
                y = s * self.stripe_h + yy
                rows.append(y if (s < self.n_stripes and y < self.h) else -1)
        return rows


def stripe_plan(h, world, rank, stripe_h=16):
    if stripe_h <= 0 or stripe_h % 8:
        raise ValueError("stripe_h must be a positive multiple of 8 (the wave tile height)")
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    n_stripes = (h + stripe_h - 1) // stripe_h
    max_local = (n_stripes + world - 1) // world
    return StripePlan(h, world, rank, stripe_h, n_stripes, max_local)


def gather_index(plan, device):
    """For the concatenation [rank0 rows | rank1 rows | ...] of padded per-rank buffers: the position
    of every image row, as an int64 tensor of length h (used with index_select)."""
    pos = [0] * plan.h
    for r in range(plan.world):
        for local, y in enumerate(plan.local_to_global_rows(r)):
            if y >= 0:
                pos[y] = r * plan.padded_rows + local
    return torch.tensor(pos, dtype=torch.int64, device=device)


def gather_rows(local, plan, index=None, group=None):
    """local: [padded_rows, row_elems...] tensor of this rank -> [h, row_elems...] on every rank."""
    if local.shape[0] != plan.padded_rows:
        raise ValueError(f"local buffer has {local.shape[0]} rows, plan needs {plan.padded_rows}")
    if plan.world == 1 and not _group_up():
        full = local
    else:                                   # a process group of one rank still goes through the collective
        full = torch.empty((plan.world * plan.padded_rows,) + tuple(local.shape[1:]), dtype=local.dtype,
                           device=local.device)
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    if index is None:
        index = gather_index(plan, local.device)
    return full.index_select(0, index)


def _group_up():
    return dist.is_available() and dist.is_initialized()


def gather_rows_to(local, plan, dst=0, index=None, group=None):
    """Like gather_rows, but only rank `dst` receives the frame ([h, row_elems...]; None elsewhere): every other rank
    sends its stripes once, over its own direct xGMI link to dst, instead of the N-1 hops of a ring all-gather.
    This is the exchange the reference's presentation needs -- one process shows the frame."""
    if local.shape[0] != plan.padded_rows:
        raise ValueError(f"local buffer has {local.shape[0]} rows, plan needs {plan.padded_rows}")
    if plan.world == 1 and not _group_up():
        full = local
    else:
        local = local.contiguous()
        full, parts = None, None
        if plan.rank == dst:
            full = torch.empty((plan.world * plan.padded_rows,) + tuple(local.shape[1:]), dtype=local.dtype,
                               device=local.device)
            parts = list(full.chunk(plan.world, dim=0))             # contiguous views: received in place
        dist.gather(local, parts, dst=dst, group=group)
        if plan.rank != dst:
            return None
    if index is None:
        index = gather_index(plan, local.device)
    return full.index_select(0, index)


class FrameGather:
    """gather_rows_to split in two, so that the exchange of frame k runs while frame k+1 is being traced.

    start(local) issues the gather as an asynchronous collective (on the "nccl" backend RCCL runs it on its own stream,
    behind everything already queued on the current stream) and returns a ticket at once; finish(ticket) makes the current
    stream wait for it and de-interleaves on rank `dst`.  Two receive buffers alternate, and the caller alternates two
    `local` buffers the same way: buffer k % 2 may be written again once finish() of frame k has been CALLED (stream
    order does the rest).  The intended order per frame: start(k), finish(k - 1), then the kernels of frame k + 1."""

    def __init__(self, plan, row_shape, dtype, device, dst=0, index=None, group=None):
        self.plan, self.dst, self.group = plan, dst, group
        self.index = gather_index(plan, device) if index is None else index
        self.collective = not (plan.world == 1 and not _group_up())
        self.full = [None, None]
        if self.collective and plan.rank == dst:
            shape = (plan.world * plan.padded_rows,) + tuple(row_shape)
            self.full = [torch.empty(shape, dtype=dtype, device=device) for _ in range(2)]
        self.k = 0

    def start(self, local):
        if local.shape[0] != self.plan.padded_rows:
            raise ValueError(f"local buffer has {local.shape[0]} rows, plan needs {self.plan.padded_rows}")
        work, full = None, local
        if self.collective:
            local = local.contiguous()
            full = self.full[self.k % 2]
            parts = list(full.chunk(self.plan.world, dim=0)) if self.plan.rank == self.dst else None
            work = dist.gather(local, parts, dst=self.dst, group=self.group, async_op=True)
        self.k += 1
        return (work, full)

    def finish(self, ticket):
        """-> [h, row_elems...] on rank dst, None elsewhere."""
        work, full = ticket
        if work is not None:
            work.wait()                              # device tensors: the current stream waits, the host does not
        if self.plan.rank != self.dst:
            return None
        return full.index_select(0, self.index)


def init_group_quiet(backend, device=None):
    """init_process_group + a first barrier with file descriptor 1 pointing at stderr meanwhile: RCCL and gloo announce
    themselves on STDOUT when their communicators come up ("RCCL version : ...", "[Gloo] Rank 0 is connected to ..."), and the
    launch scripts keep stdout for rank 0's one JSON line."""
    import os
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        dist.barrier()
        if backend == "nccl":
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def shard_range(n, world, rank):
    """Contiguous range [first, first+count) of rank; every rank's padded count is ceil(n/world)."""
    per = (n + world - 1) // world
    first = min(rank * per, n)
    return first, min(per, n - first), per


def gather_ranges(local, n, world, group=None):
    """local: [per, ...] (padded) -> [n, ...] in ray order on every rank."""
    if world == 1 and not _group_up():
        return local[:n]
    full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full[:n]
