#!/usr/bin/env python3
"""bench.py — throughput of the dsp_icpc hot path on N MI355X GPUs of one node.

A "step" is one pass of the fused dsp_icpc chain (reference src/dsp_icpc.jl:62-230,
full chain incl. CUSP/ZAC + extremestats + Intersect family = BASELINE config 3)
over one batch of synthetic HPGe traces that is already resident in HBM.
  N = 1 : 1 M x 8192-sample float32 traces on one GPU (config 3).
  N > 1 : each rank owns its own 1 M-trace shard (weak scaling), runs the same
          kernel and the [n,48] output shards are gathered to rank 0 with one
          RCCL gather per batch inside the timed region (config 4 shape); the
          gather of batch k overlaps the kernel of batch k+1 (double-buffered
          tables, all gathers complete before the closing barrier).
Prints ONE JSON line (rank 0).  `--workload pz_trap` times BASELINE config 2
(blmean -> shift -> InvCR -> Trap(10us,4us) -> max) instead; `--workload sipm` times BASELINE
config 5's shape (fused dsp_sipm, 16384-sample traces, 625 k per GPU by default).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import legenddsp_jl_amd as ldsp  # noqa: E402
from legenddsp_jl_amd import dist as ldist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md chip table)


def measured_traffic(kernel, n, L):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command
    (profiles/rNN_hbm_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
    --pmc runs, tools/profile_round.sh).  None when no pass at this batch shape is on file."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("n_traces") != n or rec.get("L") != L:
            continue
        for name, v in rec.get("kernels", {}).items():
            if name.startswith("ldsp::" + kernel + "<"):
                return {"bytes": v["hbm_bytes_per_launch_corrected"], "source": os.path.relpath(path, ROOT)}
    return None


def bench_sipm(args, n, L, world, rank, dev):
    """dsp_sipm (reference src/dsp_sipm.jl:47-158): weak scaling, each rank its own shard; the ragged trigger columns
    stay on the rank that produced them (fixed-capacity slabs + counts), scalar columns are gathered to rank 0."""
    params = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, L, 0.0, 16.0)
    wf = torch.empty((n, L), dtype=torch.float32, device=dev)
    ldsp.synth.sipm_batch(n, L, device=dev, out=wf)
    ctx = ldsp.Context(dev.index)
    ctx.enable_timing(True)
    bufs = ldsp.sipm_run(wf, params, ctx)      # allocates the output buffers once

    def step():
        ldsp.sipm_run(wf, params, ctx, out=bufs)
        if world > 1:
            ldist.gather_table(bufs[0].t().contiguous(), n * world, dst=0)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kms = []
    for _ in range(3):
        ldsp.sipm_run(wf, params, ctx, out=bufs)
        kms.append(ctx.last_kernel_ms())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        kms = sum(kms) / len(kms)
        elems = sum(int(bufs[1][g]["count"].clamp(max=ldsp._abi.LDSP_MAX_TRIG).sum()) for g in bufs[1]) * 4   # 4 ragged fields per group
        bytes_per_trace = 4 * L + 4 * 20 + 8.0 * elems / n      # measured ragged element count
        achieved = n * bytes_per_trace / (kms * 1e-3) / 1e9
        wps = n * world * args.steps / elapsed
        res = {
            "metric": "waveforms/s, fused dsp_sipm, 16384-sample f32", "value": wps, "unit": "waveforms/s",
            "msamples_per_s": wps * L / 1e6, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 5 shape: dsp_sipm, 16384-sample f32 traces", "traces_per_gpu": n, "samples": L,
                       "dt_ns": 16.0, "dsp_config": "reference test/test_dsp_sipm.jl:38-68 + sg.wl = 200 ns",
                       "ragged_elements_per_trace": elems / n},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_sipm_s4", "kernel_ms": kms, "algorithmic_bytes_per_trace": bytes_per_trace},
        }
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=None, help="traces per GPU (default 1 M; 625 k for sipm)")
    ap.add_argument("--L", type=int, default=None, help="samples per trace (default 8192; 16384 for sipm)")
    ap.add_argument("--workload", choices=["icpc", "pz_trap", "sipm"], default="icpc")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="traces timed on the host cores (0 = skip); ~10 s on 16 cores")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n = args.n if args.n is not None else (625_000 if args.workload == "sipm" else 1_000_000)
    L = args.L if args.L is not None else (16384 if args.workload == "sipm" else 8192)
    if args.workload == "sipm":
        return bench_sipm(args, n, L, world, rank, dev)
    dt = 16.0 * (8192 / L) if L < 8192 else 16.0  # 4096-sample plumbing config needs 32 ns
    cfg = ldsp.reference_test_icpc_config() if L >= 8192 else ldsp.plumbing_icpc_config_4096()
    params = ldsp.lower_icpc(cfg, 500 * ldsp.us, {}, L, 0.0, dt)

    # synthetic input, generated directly in HBM (excluded from timing)
    wf = torch.empty((n, L), dtype=torch.float32, device=dev)
    ldsp.synth.hpge_batch(n, L, device=dev, out=wf, first_trace=rank * n)
    ctx = ldsp.Context(dev.index)
    ctx.enable_timing(True)
    ncol = len(ldsp._abi.ICPC_COLS)
    out = torch.empty((n, ncol), dtype=torch.float32, device=dev) if args.workload == "icpc" else \
        torch.empty((2, n), dtype=torch.float32, device=dev)

    # N > 1: the table of batch k travels to rank 0 (RCCL, its own stream) while the kernel of batch k+1 runs: two output
    # tables per rank, two gathered tables on rank 0, and a table is overwritten only after its gather has completed
    pipe = world > 1 and args.workload == "icpc"
    outs = [out, torch.empty_like(out)] if pipe else [out]
    gathered = [torch.empty((n * world, ncol), dtype=torch.float32, device=dev) for _ in range(2)] if (pipe and rank == 0) else [None, None]
    works = [None, None]
    count = [0]

    def step():
        if args.workload == "icpc":
            k = count[0] % len(outs)
            count[0] += 1
            if works[k] is not None:
                works[k].wait()                      # the gather that read outs[k] two batches ago
                works[k] = None
            ldsp.icpc_run(wf, params, ctx, out=outs[k])
            if pipe:
                _, works[k] = ldist.gather_table(outs[k], n * world, dst=0, out=gathered[k], async_op=True)
        else:
            ldsp.icpc_pz_trap_run(wf, params, ctx, out=out)

    def fence():
        for k in range(2):
            if works[k] is not None:
                works[k].wait()                      # every gather issued so far is inside the timed region
                works[k] = None
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms, stage_ms = [], [[], []]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if rank == 0 and world == 1:
            pass
    fence()
    elapsed = time.perf_counter() - t0
    # per-launch duration of the dominant kernel, HIP events on the launch stream
    for _ in range(3):
        if args.workload == "icpc":
            ldsp.icpc_run(wf, params, ctx, out=out)
        else:
            ldsp.icpc_pz_trap_run(wf, params, ctx, out=out)
        kernel_ms.append(ctx.last_kernel_ms())
        if args.workload == "icpc":
            stage_ms[0].append(ctx.last_stage_ms(0))
            try:   # two launches only when the CUSP/ZAC stage could not be fused
                stage_ms[1].append(ctx.last_stage_ms(1))
            except ldsp.LdspError:
                pass
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total = n * world * args.steps
        wps = total / elapsed
        kms = sum(kernel_ms) / len(kernel_ms)
        # one trace read (4L) + one output row written (4 bytes x 48 columns).  dsp_icpc runs as ONE
        # launch of icpc_kernel (CUSP/ZAC fused in), so the dominant kernel's algorithmic bytes are
        # the path's.  If the fused form was not applicable (two launches), icpc_kernel writes 42 columns + a
        # 16-byte hand-over record and icpc_cz_kernel reads the trace again and writes the other 6.
        chain_bytes = 4 * L + (4 * ncol if args.workload == "icpc" else 8)
        fused = args.workload == "icpc" and not stage_ms[1]
        if args.workload == "icpc":
            k1 = sum(stage_ms[0]) / len(stage_ms[0])
            k2 = sum(stage_ms[1]) / len(stage_ms[1]) if stage_ms[1] else 0.0
            bytes_per_trace = chain_bytes if fused else 4 * L + 4 * 42 + 16
            dom_ms = k1
        else:
            bytes_per_trace, dom_ms = chain_bytes, kms
        achieved = n * bytes_per_trace / (dom_ms * 1e-3) / 1e9
        dom_kernel = "icpc_kernel" if args.workload == "icpc" else "pz_trap_kernel"
        traffic = measured_traffic(dom_kernel, n, L)
        res = {
            "metric": "waveforms/s, full dsp_icpc chain, 8192-sample f32" if args.workload == "icpc"
            else "waveforms/s, pole-zero + trapezoid sub-chain, 8192-sample f32",
            "value": wps, "unit": "waveforms/s",
            "msamples_per_s": wps * L / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE config 3: {n} x {L} f32, full dsp_icpc chain" if args.workload == "icpc"
                                    else f"BASELINE config 2: {n} x {L} f32, pole-zero + trapezoid"),
                       "traces_per_gpu": n, "samples": L, "dt_ns": dt,
                       "dsp_config": "reference test/test_dsp_icpc.jl:50-161", "tau_us": 500,
                       "gather": "rccl gather of [n,48] f32 to rank 0, overlapped with the next batch's kernel (double-buffered)" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": dom_kernel,
                         "kernel_ms": dom_ms, "algorithmic_bytes_per_trace": bytes_per_trace},
        }
        if traffic:
            res["roofline"]["traffic"] = traffic["bytes"]
            res["roofline"]["traffic_source"] = traffic["source"]
        if args.workload == "icpc":
            res["roofline"]["launches"] = "1 (icpc_kernel, CUSP/ZAC fused)" if fused else "2 (icpc_kernel + icpc_cz_kernel)"
            if not fused:
                res["roofline"]["chain"] = {  # both kernels together against the path's algorithmic bytes
                    "kernels_ms": {"icpc_kernel": k1, "icpc_cz_kernel": k2}, "algorithmic_bytes_per_trace": chain_bytes,
                    "achieved": n * chain_bytes / (kms * 1e-3) / 1e9, "frac": n * chain_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and args.cpu_sample > 0 and args.workload == "icpc":
            from oracle import oracle as orc  # checker / CPU baseline only
            orc.build()
            m = min(args.cpu_sample, n)
            host = wf[:m].cpu().numpy()
            cores = min(os.cpu_count() or 1, 16)
            orc.dsp_icpc(host[:cores], params, nthreads=cores)  # warm-up (thread pool, LSQ bases)
            t1 = time.perf_counter()
            orc.dsp_icpc(host, params, nthreads=cores)
            cpu_t = time.perf_counter() - t1
            # the reference itself is single-threaded: the same restatement on one core, smaller sample
            m1 = min(2048, m)
            t2 = time.perf_counter()
            orc.dsp_icpc(host[:m1], params, nthreads=1)
            cpu_t1 = time.perf_counter() - t2
            res["cpu_baseline"] = {
                "value": m / cpu_t, "unit": "waveforms/s", "cores": cores, "kind": "port",
                "single_thread": {"value": m1 / cpu_t1, "unit": "waveforms/s", "cores": 1, "sample": f"first {m1} traces"},
                "sample": f"first {m} traces of the same batch, float64 CPU restatement (oracle/ldsp_oracle.c), "
                          f"OpenMP over traces, direct-form CUSP/ZAC; proxy for the single-threaded Julia reference",
            }
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
