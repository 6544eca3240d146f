"""N > 1 path on CPU: world_size-2 gloo ranks shard a batch and gather the table."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from legenddsp_jl_amd import dist as ldist


def test_shard_range_partitions():
    for n in (0, 1, 7, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            r = [ldist.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ldist.shard_range(n, world, rank)
    # stand-in for the per-rank kernel output: row i of the table = f(global trace index)
    idx = torch.arange(lo, hi, dtype=torch.float32)
    tab = torch.stack([idx * (c + 1) for c in range(48)], dim=1)
    full = ldist.gather_table(tab, n, dst=0)
    # second round into a preallocated buffer (what bench.py does every step): equal shards -> the buffer itself
    nmax = -(-n // world)
    pre = torch.full((world * nmax, 48), -1.0) if rank == 0 else None
    again = ldist.gather_table(tab, n, dst=0, out=pre)
    if n == world * nmax:   # async form (bench.py's pipeline): table and work handle, complete after wait()
        pre2 = torch.full((world * nmax, 48), -2.0) if rank == 0 else None
        tab3, work = ldist.gather_table(tab, n, dst=0, out=pre2, async_op=True)
        work.wait()
        if rank == 0:
            assert torch.equal(tab3, full)
        else:
            assert tab3 is None
    if rank == 0:
        assert torch.equal(again, full)
        if n == world * nmax:
            assert again.data_ptr() == pre.data_ptr()
        q.put(full.numpy())
    else:
        assert full is None
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [10, 11])
def test_gather_table_gloo(n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = np.stack([np.arange(n, dtype=np.float32) * (c + 1) for c in range(48)], axis=1)
    np.testing.assert_array_equal(full, expect)


def _pipeline_worker(rank, world, port, q):
    """bench.py's N > 1 loop in miniature: two output tables, two gathered tables, a table is rewritten only after the
    gather that read it has completed; all gathers complete before the closing barrier."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 6
    outs = [torch.empty((n, 48)), torch.empty((n, 48))]
    gathered = [torch.empty((n * world, 48)) for _ in range(2)] if rank == 0 else [None, None]
    works, seen = [None, None], []
    for step in range(5):
        k = step % 2
        if works[k] is not None:
            works[k].wait(); works[k] = None
            if rank == 0:
                seen.append(gathered[k].clone())
        outs[k].copy_(torch.full((n, 48), float(100 * step + rank)))        # "the kernel of this batch"
        _, works[k] = ldist.gather_table(outs[k], n * world, dst=0, out=gathered[k], async_op=True)
    for k in ((5 % 2), (5 % 2) ^ 1):     # oldest first
        if works[k] is not None:
            works[k].wait()
            if rank == 0:
                seen.append(gathered[k].clone())
    dist.barrier()
    if rank == 0:
        q.put([t.numpy() for t in seen])
    dist.destroy_process_group()


def test_overlapped_gather_pipeline_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    seen = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(seen) == 5
    for step, tab in enumerate(seen):     # every batch arrived intact, in order, rank blocks in place
        assert (tab[:6] == 100 * step).all() and (tab[6:] == 100 * step + 1).all(), step
