"""RCCL smoke check on the one-GPU box: the collectives `legenddsp_jl_amd.dist` issues at N > 1 (gather of the [n, 48] table into
row blocks of the result — blocking and async_op —, the MIN all-reduce of the argument check, an int64 gather of the counts),
issued with the same argument shapes on a one-rank `nccl` group.  `gather_table` / `gather_ragged` return early at world size 1,
so the calls are made directly; what this shows is that the process group comes up on this image (device_id form, dmabuf IPC
setting) and that RCCL takes these tensors and orders the async gather against the compute stream.  Point-to-point payloads need a
second GPU and are covered by the gloo tests only.
Usage (GPU box): python tools/rccl_world1_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import legenddsp_jl_amd as ldsp

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, L = 4096, 8192
p = ldsp.lower_icpc(ldsp.reference_test_icpc_config(), 500 * ldsp.us, {}, L, 0.0, 16.0)
wf = ldsp.synth.hpge_batch(n, L, device=dev)
tab = ldsp.icpc_run(wf, p)
out = torch.empty_like(tab)
bufs = list(out.split(n, dim=0))
dist.gather(tab.contiguous(), bufs, dst=0)
torch.cuda.synchronize()
assert torch.equal(out.view(torch.int32), tab.view(torch.int32)), "blocking gather"
out.zero_()
work = dist.gather(tab.contiguous(), bufs, dst=0, async_op=True)
tab2 = ldsp.icpc_run(wf, p)          # the next batch's kernel runs while the gather is in flight
work.wait()
torch.cuda.synchronize()
assert torch.equal(out.view(torch.int32), tab.view(torch.int32)) and torch.equal(tab2.view(torch.int32), tab.view(torch.int32)), "async gather"
ok = torch.tensor([1], dtype=torch.int32, device=dev)
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
assert int(ok.item()) == 1
cnt = torch.arange(n, dtype=torch.int64, device=dev)[:, None].contiguous()
cout = torch.empty_like(cnt)
dist.gather(cnt, list(cout.split(n, dim=0)), dst=0)
torch.cuda.synchronize()
assert torch.equal(cout, cnt)
dist.barrier()
dist.destroy_process_group()
print(f"rccl world-1 check ok: nccl group up (torch {torch.__version__}), gather [{n}, {tab.shape[1]}] f32 blocking + async_op, all_reduce MIN int32, gather int64")
