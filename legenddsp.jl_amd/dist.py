"""Multi-GPU driver of the hot path: shard the waveform batch, gather the table.

The path shards naturally: no statement of dsp_icpc / dsp_sipm
combines data across traces.  One process per GPU; rank r owns the contiguous
trace range `shard_range(n, world, r)`, runs the fused kernel on it, and the
only collective is ONE gather of the [n_r, 48] float32 output shards to rank 0
(`torch.distributed` backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  No data-path collective.  The gather of batch k can run (async, on the collective's stream) while the
kernel of batch k+1 computes: `gather_table(..., async_op=True)` with double-buffered tables, as bench.py does.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, world: int, rank: int):
    """Contiguous, balanced partition of range(n): first (n % world) ranks get one extra trace."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_table(tab: torch.Tensor, n_total: int, dst: int = 0, group=None, out: torch.Tensor = None, async_op: bool = False):
    """Gather the per-rank [n_r, C] shards into the [n_total, C] table on `dst`
    (row order = global trace index).  Returns the table on dst, None elsewhere.

    Shards may differ by one row; they are padded to the largest shard so that a
    single fixed-size gather moves everything (one collective, direct peer links).
    `out` (dst only, optional): a preallocated [world * ceil(n_total / world), C] buffer that receives the
    shards in place — with equal shards the result is a view of it (no allocation, no copy).
    `async_op=True` (equal shards only): returns `(table_or_None, work)` at once; the transfer runs on the collective's own
    stream — concurrently with whatever the caller launches next on the compute stream — and `work.wait()` orders the compute
    stream behind it.  The caller must not overwrite `tab` (nor read the table) before that."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tab
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    C = tab.shape[1]
    nmax = -(-n_total // world)
    send = tab
    if tab.shape[0] != nmax:
        send = torch.zeros((nmax, C), dtype=tab.dtype, device=tab.device)
        send[: tab.shape[0]] = tab
    bufs = None
    if rank == dst:
        if out is None or out.shape != (world * nmax, C) or out.dtype != tab.dtype or out.device != tab.device:
            out = torch.empty((world * nmax, C), dtype=tab.dtype, device=tab.device)
        bufs = list(out.split(nmax, dim=0))   # contiguous row blocks of the result: the gather writes the table itself
    if async_op:
        if n_total != world * nmax:
            raise ValueError("async gather needs equal shards")
        work = dist.gather(send.contiguous(), bufs, dst=dst, group=group, async_op=True)
        return (out if rank == dst else None), work
    dist.gather(send.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if n_total == world * nmax:
        return out
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, world, r)
        parts.append(bufs[r][: hi - lo])
    return torch.cat(parts, dim=0)


def gather_ragged(values: torch.Tensor, counts: torch.Tensor, n_total: int, dst: int = 0, group=None):
    """Gather one ragged column family (counts -> exclusive scan on root -> payload).

    `counts` [n_r]: elements per trace of this rank's shard (trace order); `values` [sum(counts), F]: the shard's
    compacted elements, F fields per element (e.g. x, x_high, x_tot, max of one IntersectMaximum call — the
    `VectorOfVectors` columns of reference src/dsp_sipm.jl:149-156 share their offsets per trigger group).
    Returns `(offsets [n_total + 1] int64, values [sum over ranks, F])` on `dst` — `values[offsets[i]:offsets[i+1]]`
    is global trace i — and None elsewhere.

    Three steps: (1) the per-trace counts travel as a one-column table (`gather_table`, fixed size); (2) the root
    scans them: offsets of every trace, and from the shard ranges the element total and start of every peer;
    (3) the payloads, whose sizes differ per peer, travel as point-to-point transfers straight into their place in
    the result (RCCL: one ncclGroup of send/recv over the direct xGMI links; gloo in the CPU tests).  A peer with no
    elements sends nothing and the root posts no receive for it: both sides decide from the same counts (the peer's own sum,
    the root's scan of the gathered table), so the point-to-point operations always pair up — also when several peers are empty."""
    if values.dim() == 1:
        values = values[:, None]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        off = torch.zeros(len(counts) + 1, dtype=torch.int64, device=counts.device)
        off[1:] = torch.cumsum(counts.to(torch.int64), 0)
        return off, values
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # Argument check BEFORE any data moves, and agreed on by every rank: a rank that raised on its own would leave the others
    # blocked in the gather below for ever (and a payload that does not match its counts would desynchronise the transfers).
    lo_, hi_ = shard_range(n_total, world, rank)
    ok = torch.tensor([1 if (int(values.shape[0]) == int(counts.sum()) and int(counts.shape[0]) == hi_ - lo_) else 0],
                      dtype=torch.int32, device=counts.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        raise ValueError("gather_ragged: on some rank `values` does not hold exactly sum(counts) rows, or `counts` is not that "
                         "rank's shard of n_total traces (checked on every rank: all of them raise)")
    cnt = gather_table(counts.to(torch.int64)[:, None].contiguous(), n_total, dst=dst, group=group)
    if rank != dst:
        if values.shape[0] > 0:
            dist.batch_isend_irecv([dist.P2POp(dist.isend, values.contiguous(), dst, group)])[-1].wait()
        return None
    cnt = cnt[:, 0]
    offsets = torch.zeros(n_total + 1, dtype=torch.int64, device=cnt.device)
    offsets[1:] = torch.cumsum(cnt, 0)
    bounds = [shard_range(n_total, world, r) for r in range(world)]
    starts = offsets[torch.tensor([b[0] for b in bounds] + [n_total], device=offsets.device)].tolist()   # one host read
    out = torch.empty((starts[-1], values.shape[1]), dtype=values.dtype, device=values.device)
    ops = []
    for r in range(world):
        lo, hi = starts[r], starts[r + 1]
        if r == rank:
            out[lo:hi] = values
        elif hi > lo:
            ops.append(dist.P2POp(dist.irecv, out[lo:hi], r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return offsets, out
