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


def _ragged_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ldist.shard_range(n, world, rank)
    idx = torch.arange(lo, hi)
    # unequal per-rank trigger totals: trace i has (i * 7) % 5 elements on rank 0's range and three times as many on
    # rank 1's; one variant (n = 9) leaves rank 1 with traces that all have zero elements
    cnt = (idx * 7) % 5 * (1 + 2 * rank)
    if n == 9 and rank == 1:
        cnt = torch.zeros_like(cnt)
    vals = torch.repeat_interleave(idx, cnt).to(torch.float32)
    payload = torch.stack([vals, vals * 2 + 1, -vals], dim=1)      # three fields per element
    res = ldist.gather_ragged(payload, cnt.to(torch.int32), n, dst=0)
    if rank == 0:
        off, out = res
        q.put((off.numpy(), out.numpy()))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [9, 10, 11])
def test_gather_ragged_gloo(n):
    """counts -> exclusive scan on root -> payload with per-peer sizes (reference columns
    src/dsp_sipm.jl:149-156), world 2, unequal per-rank totals, unequal shards (n odd), an empty peer (n = 9)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    off, out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    lo1 = ldist.shard_range(n, 2, 1)[0]
    idx = np.arange(n)
    cnt = (idx * 7) % 5 * np.where(idx >= lo1, 3, 1)
    if n == 9:
        cnt[lo1:] = 0
    assert off[0] == 0 and np.array_equal(np.diff(off), cnt)
    vals = np.repeat(idx, cnt).astype(np.float32)
    np.testing.assert_array_equal(out, np.stack([vals, vals * 2 + 1, -vals], axis=1))
    for i in range(n):       # element i of the VectorOfVectors is trace i's
        assert (out[off[i]:off[i + 1], 0] == i).all()


def _ragged_worker4(rank, world, port, n, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ldist.shard_range(n, world, rank)
    idx = torch.arange(lo, hi)
    cnt = torch.where(torch.tensor(rank in (0, 3)), (idx * 3) % 4 + 1, torch.zeros_like(idx))   # ranks 1 and 2: no element at all
    vals = torch.repeat_interleave(idx, cnt).to(torch.float32)[:, None]
    if mode == "bad" and rank == 2:
        vals = torch.zeros((3, 1))          # a payload that does not match its counts, on ONE rank
    try:
        res = ldist.gather_ragged(vals, cnt.to(torch.int32), n, dst=0)
        if rank == 0:
            q.put(("ok", res[0].numpy(), res[1].numpy()))
    except ValueError:
        q.put(("raised", rank, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["empty_peers", "bad"])
def test_gather_ragged_world4_empty_peers_and_collective_check(mode):
    """world 4 (gloo): two of the three peers have no element — the root posts no receive for them and they send nothing; and a
    payload that does not match its counts on ONE rank makes EVERY rank raise (the check is an all-reduce in front of the
    transfers: nobody is left waiting in the gather)."""
    n, world = 22, 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker4, args=(r, world, port, n, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(1 if mode == "empty_peers" else world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if mode == "bad":
        assert sorted(g[1] for g in got) == [0, 1, 2, 3] and all(g[0] == "raised" for g in got)
        return
    _, off, out = got[0]
    idx = np.arange(n)
    owner = np.array([next(r for r in range(world) if ldist.shard_range(n, world, r)[0] <= i < ldist.shard_range(n, world, r)[1]) for i in idx])
    cnt = np.where(np.isin(owner, (0, 3)), (idx * 3) % 4 + 1, 0)
    assert np.array_equal(np.diff(off), cnt)
    np.testing.assert_array_equal(out[:, 0], np.repeat(idx, cnt).astype(np.float32))


def test_gather_ragged_single_process():
    cnt = torch.tensor([2, 0, 3], dtype=torch.int32)
    vals = torch.arange(5, dtype=torch.float32)
    off, out = ldist.gather_ragged(vals, cnt, 3)
    assert off.tolist() == [0, 2, 2, 5] and out[:, 0].tolist() == [0, 1, 2, 3, 4]


@pytest.mark.parametrize("workload", ["icpc", "sipm"])
def test_bench_self_launch_two_ranks(workload):
    """`python bench.py --gpus 2` with no rank environment starts its own two ranks (the driver's command); --dry-run
    runs the launcher, the process group and every gather of the real loop on CPU tensors and prints ONE JSON line."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--n", "500", "--steps", "3", "--warmup", "2",
                        "--workload", workload], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["gather_ok"] is True


def test_bench_rehearsal_single_batch_under_the_drivers_launcher():
    """One batch in all (`--steps 1 --warmup 0`: the second table of the overlapped gather is never used), started the way the driver starts
    N > 1 — `python -m torch.distributed.run` —, with the traces given as `--traces` (the launcher's parser claims a script option spelled
    `--n` as an abbreviation of its own `--nnodes` / `--nproc-per-node` and exits)."""
    import json, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--traces", "203",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["gather_ok"] is True


def test_bench_launcher_relays_failure():
    """a rank that exits non-zero (here: argparse rejects the workload in every child) makes the launcher exit non-zero"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "nope"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0


@pytest.mark.parametrize("workload", ["icpc", "sipm"])
def test_bench_rehearsal_at_the_real_width(workload):
    """The width the driver's scaling run uses: eight ranks of one node, self-launched (`python bench.py --gpus 8`), both workloads —
    launcher, process group, contiguous shards, the table gather and (sipm) the ragged gather with its count / scan / payload passes
    (src/dsp_sipm.jl:149-156 are the ragged columns that travel), on CPU tensors."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--n", "203", "--steps", "2", "--warmup", "1",
                        "--workload", workload], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["dry_run"] is True and rec["gather_ok"] is True


def test_bench_rejects_a_local_rank_without_a_device():
    """LOCAL_RANK is the device index: a rank whose LOCAL_RANK has no GPU behind it (here: a box without GPUs) stops with a message
    that names both numbers instead of failing inside set_device."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_PORT")}, LOCAL_RANK="3", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-secondary", "--cpu-sample", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "LOCAL_RANK=3" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])
