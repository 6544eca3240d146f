#!/usr/bin/env python3
"""bench.py — throughput of the dsp_icpc hot path on N MI355X GPUs of one node.

A "step" is one pass of the fused dsp_icpc chain (reference src/dsp_icpc.jl:62-230,
full chain incl. CUSP/ZAC + extremestats + Intersect family = BASELINE config 3)
over one batch of synthetic HPGe traces that is already resident in HBM.
  N = 1 : 1 M x 8192-sample float32 traces on one GPU (config 3).  The default run also times, after the
          headline, BASELINE config 2 (pole-zero + trapezoid sub-chain, same batch) and config 5's single-GPU
          shard (fused dsp_sipm, 625 k x 16384) and reports them under "secondary" (`--no-secondary` skips them).
  N > 1 : each rank owns its own shard (weak scaling: 1.25 M traces per rank = config 4's 10 M over 8), runs
          the same kernel and the [n,48] output shards are gathered to rank 0 with one RCCL gather per batch
          inside the timed region; the gather of batch k overlaps the kernel of batch k+1 (double-buffered
          tables, all gathers complete before the closing barrier).  `--workload sipm` (config 5: 625 k x 16384
          per rank) gathers the 20 scalar columns AND the 12 ragged trigger columns (counts -> exclusive scan on
          the root -> payload with per-peer sizes, `dist.gather_ragged`).
Launch: `python bench.py --gpus N ...` starts its own N ranks (one fresh process per GPU; the parent touches
neither torch nor the GPU and relays the children's exit status); under `python -m torch.distributed.run` (RANK /
WORLD_SIZE already set) it is one of the ranks.  Prints ONE JSON line (rank 0).
`--dry-run` rehearses the launch + gather plumbing on CPU (gloo, kernels replaced by table fills, no number of
merit): what the CPU tests run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md chip table)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--traces", dest="n", type=int, default=None, help="traces per GPU; spell it --traces under torch.distributed.run, whose parser claims --n (default: icpc 1 M at N = 1, 1.25 M at N > 1; sipm 625 k)")
    ap.add_argument("--L", type=int, default=None, help="samples per trace (default 8192; 16384 for sipm)")
    ap.add_argument("--workload", choices=["icpc", "pz_trap", "sipm"], default="icpc")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="traces timed on the host cores (0 = skip); ~10 s on 16 cores")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: do not append the config 2 / config 5 lines")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the launcher and the gathers (gloo; no kernels, no GPU)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no rank environment -> N child processes, one per GPU

def launch_ranks(args, argv):
    """Start N fresh rank processes of this script and wait for them.  The parent imports neither torch nor the
    package (it must not initialise the GPU); it relays a non-zero exit status and ends the other ranks when one dies."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    alive = set(range(len(procs)))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # a dead rank leaves the others in a collective for ever: end exactly those processes
                    procs[q].terminate()
        if alive:
            time.sleep(0.2)
    return rc if rc >= 0 else 1


# ------------------------------------------------------------------------------------------------------------------

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
        recs = rec if isinstance(rec, list) else [rec]
        for rec in recs:
            if rec.get("n_traces") != n or rec.get("L") != L:
                continue
            for name, v in rec.get("kernels", {}).items():
                if name.startswith("ldsp::" + kernel + "<"):
                    return {"bytes": v["hbm_bytes_per_launch_corrected"], "source": os.path.relpath(path, ROOT)}
    return None


class Env:
    """What a rank needs: world / rank, its device, torch.distributed started (N > 1)."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dry = args.dry_run
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        if self.dry:
            self.dev = torch.device("cpu")
        else:
            ndev = torch.cuda.device_count()     # (counting devices does not initialise one)
            if not 0 <= self.local_rank < ndev:
                raise SystemExit(f"bench.py: rank {self.rank} has LOCAL_RANK={self.local_rank} but this process sees {ndev} GPU(s) "
                                 "(one rank per GPU of ONE node: check --gpus / --nproc-per-node against the visible devices)")
            self.dev = torch.device("cuda", self.local_rank)
            torch.cuda.set_device(self.dev)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.dry:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", device_id=self.dev)    # "nccl" is RCCL on ROCm

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize(self.dev)

    def fence(self):
        self.sync()
        if self.world > 1:
            self.dist.barrier()
            self.sync()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


PREWARM_S = 0.15   # untimed launches in front of the W warm-up steps, see timed()


def timed(env, args, step, finish=lambda: None):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.
    In front of the W warm-up steps the step is repeated (untimed) until 150 ms have passed: after an idle period — the CPU-baseline leg
    of the previous workload takes seconds — this GPU takes tens of milliseconds of back-to-back work to return to its full clocks, and
    W = 2 launches of a 6 ms kernel are over before it has (profiles/r04_pz_trap_launch_spread.txt: 24 launches back to back 5.68–5.79 ms,
    each behind 20 ms of idle 6.8–7.2 ms; round 3's unexplained 5.6–6.7 ms spread of config 2)."""
    if not env.dry and env.world == 1:   # (one rank only: a time-based count would differ between ranks, and a step of N > 1 holds a collective)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < PREWARM_S:
            step()
            finish()
            env.sync()
    for _ in range(args.warmup):
        step()
    finish()
    env.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()
    env.fence()
    return env.max_over_ranks(time.perf_counter() - t0)


# Instruction-issue ceilings of the full dsp_icpc chain at L = 8192 (waveforms/s per GPU).  Round 4 measured what
# binds the kernel: instruction ISSUE of every kind (profiles/r04_lean3_whatif.txt: 832 extra v_fmac per wave cost 1.8 cycles each, 832
# extra s_add 2.0, 208 extra ds_read_b32 4.5 — their full price, nothing hides; removing all twelve barriers of the CUSP / ZAC stage buys
# 2.6 %), and that SQ_ACTIVE_INST_VALU — round 3's "VALU 90 % busy" — is an instruction count (profiles/r04_micro_valu_cost.txt).
#   "algorithmic": the minimum arithmetic of the restructured chain (~85 VALU operations per sample), perfectly packed at 2 cycles per
#   instruction and SIMD, no scalar, LDS or barrier time;
#   "instruction_stream": every instruction the shipped kernel issues (SQ_INSTS per wave, all kinds, profiles/r04_lean3_phase_insts.txt)
#   at the cheapest issue price measured on this chip (2.08 cycles per instruction and SIMD, v_fmac at >= 4 waves per SIMD):
#   2.4e9 * 1024 SIMDs / (8 waves * INSTS_PER_WAVE * 2.08).  The kernel cannot exceed it without issuing fewer instructions.
INSTS_PER_WAVE_ICPC = 5794
ISSUE_CEILING_ICPC = {"algorithmic": 65.0e6, "instruction_stream": 2.4e9 * 1024 / (8 * INSTS_PER_WAVE_ICPC * 2.08)}


def roofline(achieved_gbs, kernel, kernel_ms, bytes_per_trace, traffic, issue_of=None):
    r = {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
         "traffic": None, "kernel": kernel, "kernel_ms": kernel_ms, "algorithmic_bytes_per_trace": bytes_per_trace}
    if traffic:
        r["traffic"] = traffic["bytes"]
        r["traffic_source"] = traffic["source"]
    if issue_of is not None:   # the second ceiling of this chain: instruction issue (waveforms/s of ONE GPU against it)
        r["issue"] = {"unit": "waveforms/s", "achieved": issue_of,
                      "ceiling_algorithmic": ISSUE_CEILING_ICPC["algorithmic"], "frac_algorithmic": issue_of / ISSUE_CEILING_ICPC["algorithmic"],
                      "ceiling_instruction_stream": ISSUE_CEILING_ICPC["instruction_stream"],
                      "frac_instruction_stream": issue_of / ISSUE_CEILING_ICPC["instruction_stream"]}
    return r


# ------------------------------------------------------------------------------------------------------------------
# dsp_icpc (config 3 / 4) and the pole-zero + trapezoid sub-chain (config 2)

def bench_icpc(env, args, n, L, workload, wf=None, pars_filter=None, label=None):
    torch = env.torch
    from legenddsp_jl_amd import dist as ldist
    world, rank, dev = env.world, env.rank, env.dev
    ncol = 48
    if not env.dry:
        import legenddsp_jl_amd as ldsp
        # the reference configuration's windows end at 110 us = sample 6875 at 16 ns: every trace of >= 7000 samples runs it unchanged;
        # shorter traces are sampled more coarsely and take the degree-2 plumbing configuration (4096 samples at 32 ns)
        ref_cfg = L >= 7000
        dt = 16.0 if ref_cfg else 16.0 * (8192 / L)
        cfg = ldsp.reference_test_icpc_config() if ref_cfg else ldsp.plumbing_icpc_config_4096()
        cfg_name = "reference test/test_dsp_icpc.jl:50-161" if ref_cfg else "plumbing_icpc_config_4096 (reference configuration with sg_flt_degree = int_interpolation_order = 2)"
        params = ldsp.lower_icpc(cfg, 500 * ldsp.us, pars_filter or {}, L, 0.0, dt)
        if wf is None:   # synthetic input, generated directly in HBM (excluded from timing)
            wf = torch.empty((n, L), dtype=torch.float32, device=dev)
            ldsp.synth.hpge_batch(n, L, device=dev, out=wf, first_trace=rank * n)
        ctx = ldsp.Context(dev.index)
        ctx.enable_timing(True)
        ncol = len(ldsp._abi.ICPC_COLS)
    else:
        dt, cfg_name = 16.0, "dry run"
    out = torch.empty((n, ncol), dtype=torch.float32, device=dev) if workload == "icpc" else \
        torch.empty((2, n), dtype=torch.float32, device=dev)

    def kernel(dst, k):
        if env.dry:     # stand-in for the kernel: row i = f(global trace index, batch)
            idx = torch.arange(rank * n, (rank + 1) * n, dtype=torch.float32)
            dst.copy_(idx[:, None] + 1000.0 * k if workload == "icpc" else idx[None, :].expand(2, n))
        elif workload == "icpc":
            ldsp.icpc_run(wf, params, ctx, out=dst)
        else:
            ldsp.icpc_pz_trap_run(wf, params, ctx, out=dst)

    # N > 1: the table of batch k travels to rank 0 (RCCL, its own stream) while the kernel of batch k+1 runs: two output
    # tables per rank, two gathered tables on rank 0, and a table is overwritten only after its gather has completed
    pipe = world > 1 and workload == "icpc"
    outs = [out, torch.empty_like(out)] if pipe else [out]
    gathered = [torch.empty((n * world, ncol), dtype=torch.float32, device=dev) for _ in range(2)] if (pipe and rank == 0) else [None, None]
    works = [None, None]
    count = [0]

    def step():
        k = count[0] % len(outs)
        if works[k] is not None:
            works[k].wait()                      # the gather that read outs[k] two batches ago
            works[k] = None
        kernel(outs[k], count[0])
        count[0] += 1
        if pipe:
            _, works[k] = ldist.gather_table(outs[k], n * world, dst=0, out=gathered[k], async_op=True)

    def finish():
        for k in range(2):
            if works[k] is not None:
                works[k].wait()                  # every gather issued so far is inside the timed region
                works[k] = None

    elapsed = timed(env, args, step, finish)
    res = None
    if env.dry:
        if rank == 0:
            ok = True
            if pipe:    # the last two batches arrived intact, rank blocks in place
                for k in range(2):
                    b = count[0] - 1 - ((count[0] - 1 - k) % 2)
                    if b < 0:      # (a single batch in all: the second table was never used)
                        continue
                    exp = torch.arange(n * world, dtype=torch.float32)[:, None] + 1000.0 * b
                    ok = ok and bool((gathered[k] == exp).all())
            res = {"metric": "dry run (launcher + gather rehearsal on CPU, no kernel)", "value": None, "unit": "waveforms/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "gather_ok": ok,
                   "config": {"workload": workload, "traces_per_gpu": n, "samples": L}}
            if not ok:
                raise SystemExit("dry run: gathered table differs from the expected rows")
        return res, None

    # per-launch duration of the dominant kernel, HIP events on the launch stream
    kernel_ms, stage_ms = [], [[], []]
    for _ in range(3):
        kernel(out, 0)
        kernel_ms.append(ctx.last_kernel_ms())
        dom_kernel = ctx.last_kernel_name()     # the name rocprofv3 reports, asked of the library
        if workload == "icpc":
            stage_ms[0].append(ctx.last_stage_ms(0))
            try:   # two launches only when the CUSP/ZAC stage could not be fused
                stage_ms[1].append(ctx.last_stage_ms(1))
            except ldsp.LdspError:
                pass
    env.sync()
    if rank == 0:
        wps = n * world * args.steps / elapsed
        kms = sum(kernel_ms) / len(kernel_ms)
        # one trace read (4L) + one output row written (4 bytes x 48 columns).  dsp_icpc runs as ONE
        # launch of icpc_kernel (CUSP/ZAC fused in), so the dominant kernel's algorithmic bytes are
        # the path's.  If the fused form was not applicable (two launches), icpc_kernel writes 42 columns + a
        # 16-byte hand-over record and icpc_cz_kernel reads the trace again and writes the other 6.
        chain_bytes = 4 * L + (4 * ncol if workload == "icpc" else 8)
        fused = workload == "icpc" and not stage_ms[1]
        if workload == "icpc":
            k1 = sum(stage_ms[0]) / len(stage_ms[0])
            k2 = sum(stage_ms[1]) / len(stage_ms[1]) if stage_ms[1] else 0.0
            bytes_per_trace = chain_bytes if fused else 4 * L + 4 * 42 + 16
            dom_ms = k1
        else:
            bytes_per_trace, dom_ms = chain_bytes, kms
        achieved = n * bytes_per_trace / (dom_ms * 1e-3) / 1e9
        res = {
            "metric": "waveforms/s, full dsp_icpc chain, 8192-sample f32" if workload == "icpc"
            else "waveforms/s, pole-zero + trapezoid sub-chain, 8192-sample f32",
            "value": wps, "unit": "waveforms/s",
            "msamples_per_s": wps * L / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE config {3 if world == 1 else 4}: {n}{' per GPU' if world > 1 else ''} x {L} f32, full dsp_icpc chain"
                                    if workload == "icpc" else f"BASELINE config 2: {n} x {L} f32, pole-zero + trapezoid"),
                       "traces_per_gpu": n, "samples": L, "dt_ns": dt,
                       "dsp_config": cfg_name, "tau_us": 500,
                       "gather": "rccl gather of [n,48] f32 to rank 0, overlapped with the next batch's kernel (double-buffered)" if pipe else "none"},
            "roofline": roofline(achieved, dom_kernel, dom_ms, bytes_per_trace, measured_traffic(dom_kernel, n, L),
                                 issue_of=(n / (dom_ms * 1e-3)) if (workload == "icpc" and L == 8192) else None),
        }
        if label:
            res["config"]["workload"] = label
        if workload == "icpc":
            res["roofline"]["launches"] = f"1 ({dom_kernel}, CUSP/ZAC fused)" if fused else "2 (icpc_kernel + icpc_cz_kernel)"
            if not fused:
                res["roofline"]["chain"] = {  # both kernels together against the path's algorithmic bytes
                    "kernels_ms": {"icpc_kernel": k1, "icpc_cz_kernel": k2}, "algorithmic_bytes_per_trace": chain_bytes,
                    "achieved": n * chain_bytes / (kms * 1e-3) / 1e9, "frac": n * chain_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and args.cpu_sample > 0 and workload == "icpc" and label is None:
            res["cpu_baseline"] = cpu_baseline(args, wf, params, n)
    return res, wf


def cpu_model():
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line.startswith("Model name"):
                return line.split(":", 1)[1].strip()
    except (OSError, subprocess.SubprocessError):
        pass
    return "unknown"


def cpu_baseline(args, wf, params, n):
    """the CPU restatement structured like the reference (one materialised pass per broadcast, float64), CUSP / ZAC by
    direct convolution AND by FFT (upstream's ConvolutionFilter has both; which one dsp_icpc takes is not visible in the
    reference), each on all host cores (one OpenMP worker per core over disjoint trace ranges) and on ONE core — the reference's
    own execution model is single-threaded.  `value` = the faster all-core form."""
    from oracle import oracle as orc  # checker / CPU baseline only
    orc.build()
    m = min(args.cpu_sample, n)
    host = wf[:m].cpu().numpy()
    cores = min(os.cpu_count() or 1, 16)
    m1 = min(2048, m)
    legs = {}
    for form, fft in (("direct", False), ("fft", True)):
        orc.set_fir_mode(fft)
        try:
            orc.dsp_icpc(host[:cores], params, nthreads=cores)  # warm-up (thread pool, LSQ bases)
            t1 = time.perf_counter()
            orc.dsp_icpc(host, params, nthreads=cores)
            t_all = time.perf_counter() - t1
            t2 = time.perf_counter()
            orc.dsp_icpc(host[:m1], params, nthreads=1)
            t_one = time.perf_counter() - t2
        finally:
            orc.set_fir_mode(False)
        legs[form] = {"all_cores": {"value": m / t_all, "unit": "waveforms/s", "cores": cores, "sample": f"first {m} traces"},
                      "single_thread": {"value": m1 / t_one, "unit": "waveforms/s", "cores": 1, "sample": f"first {m1} traces"}}
    best = max(legs, key=lambda k: legs[k]["all_cores"]["value"])
    return {
        "value": legs[best]["all_cores"]["value"], "unit": "waveforms/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "cusp_zac_form": best, "forms": legs,
        "single_thread": legs[best]["single_thread"],
        "sample": f"first {m} traces of the same batch (one core: first {m1}), float64 CPU restatement (oracle/ldsp_oracle.c) structured like the "
                  f"reference's broadcasts, OpenMP over traces; 2375-tap CUSP / ZAC by direct convolution and by FFT (complex radix-2 transform of length 16384 — not a real-input FFT "
                  f"—, the taps' transform computed once per thread and parameter set), the faster one reported; "
                  f"proxy for the single-threaded Julia reference (not runnable here: no Julia), measured, not extrapolated",
    }


# ------------------------------------------------------------------------------------------------------------------
# dsp_sipm (config 5)

SIPM_GROUPS = ("trig", "trig_DC", "trig_trap", "trig_DC_trap")
SIPM_FIELDS = ("x", "x_high", "x_tot", "max")


def bench_sipm(env, args, n, L):
    """dsp_sipm (reference src/dsp_sipm.jl:47-158): weak scaling, each rank its own shard.  N > 1: the 20 scalar columns
    travel as one table, every trigger group's ragged columns (reference :149-156) as counts + compacted payload
    (`dist.gather_ragged`), all inside the timed region."""
    torch = env.torch
    from legenddsp_jl_amd import dist as ldist
    world, rank, dev = env.world, env.rank, env.dev
    if not env.dry:
        import legenddsp_jl_amd as ldsp
        from legenddsp_jl_amd.extractors import compact_fields
        from legenddsp_jl_amd.routines import sipm_resolve_overflow
        params = ldsp.lower_sipm(ldsp.reference_test_sipm_config(), {"sg": {"wl": 200 * ldsp.ns}}, L, 0.0, 16.0)
        wf = torch.empty((n, L), dtype=torch.float32, device=dev)
        ldsp.synth.sipm_batch(n, L, device=dev, out=wf, first_trace=rank * n)
        ctx = ldsp.Context(dev.index)
        ctx.enable_timing(True)
        bufs = ldsp.sipm_run(wf, params, ctx)      # allocates the output buffers once
    gathered = {}

    def step():
        if env.dry:   # stand-in: trace i of the job has (i % 5) + g triggers in group g, element value = global trace index
            idx = torch.arange(rank * n, (rank + 1) * n)
            sc = idx[:, None].to(torch.float32).expand(n, 20).contiguous()
            rag = {}
            for g, name in enumerate(SIPM_GROUPS):
                cnt = (idx % 5) + g
                rag[name] = (torch.repeat_interleave(idx, cnt).to(torch.float32)[:, None].expand(-1, 4).contiguous(), cnt)
        else:
            sc_t, trig = ldsp.sipm_run(wf, params, ctx, out=bufs)
            if world == 1:
                return
            sc = sc_t.t().contiguous()
            trig = sipm_resolve_overflow(wf, params, ctx, trig)     # traces with more triggers than a slab holds run again
            rag = {g: compact_fields(trig[g], SIPM_FIELDS) for g in SIPM_GROUPS}
        if world > 1:
            gathered["scalars"] = ldist.gather_table(sc, n * world, dst=0)
            for g in SIPM_GROUPS:
                gathered[g] = ldist.gather_ragged(rag[g][0], rag[g][1], n * world, dst=0)

    elapsed = timed(env, args, step)
    if env.dry:
        res = None
        if rank == 0:
            ok = True
            if world > 1:
                idx = torch.arange(n * world)
                ok = bool((gathered["scalars"][:, 0] == idx.to(torch.float32)).all())
                for g, name in enumerate(SIPM_GROUPS):
                    off, val = gathered[name]
                    cnt = (idx % 5) + g
                    ok = ok and bool((off[1:] - off[:-1] == cnt).all()) and bool((val[:, 0] == torch.repeat_interleave(idx, cnt).to(torch.float32)).all())
            res = {"metric": "dry run (launcher + scalar and ragged gather rehearsal on CPU, no kernel)", "value": None, "unit": "waveforms/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True, "gather_ok": ok,
                   "config": {"workload": "sipm", "traces_per_gpu": n, "samples": L}}
            if not ok:
                raise SystemExit("dry run: gathered sipm columns differ from the expected rows")
        return res
    kms = []
    for _ in range(3):
        ldsp.sipm_run(wf, params, ctx, out=bufs)
        kms.append(ctx.last_kernel_ms())
    env.sync()
    res = None
    if rank == 0:
        kms = sum(kms) / len(kms)
        # the 12 ragged columns the table keeps (reference :149-156): x, max of the two SG groups, all four fields of the trapezoid
        # groups; positions (x, x_high, x_tot) are stored as Float64 (8 B), maxima as float32 (4 B): what the kernel writes
        bytes_per_trig = {"trig": 8 + 4, "trig_DC": 8 + 4, "trig_trap": 3 * 8 + 4, "trig_DC_trap": 3 * 8 + 4}
        per_group = {"trig": 2, "trig_DC": 2, "trig_trap": 4, "trig_DC_trap": 4}
        counts = {g: int(bufs[1][g]["count"].sum()) for g in bufs[1]}
        elems = sum(counts[g] * per_group[g] for g in counts)
        bytes_per_trace = 4 * L + 4 * 20 + sum(counts[g] * bytes_per_trig[g] for g in counts) / n
        achieved = n * bytes_per_trace / (kms * 1e-3) / 1e9
        wps = n * world * args.steps / elapsed
        res = {
            "metric": "waveforms/s, fused dsp_sipm, 16384-sample f32", "value": wps, "unit": "waveforms/s",
            "msamples_per_s": wps * L / 1e6, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 signal arithmetic, Float64 time axis and trigger positions (the reference forces Float64 throughout, src/dsp_sipm.jl:87-88)", "data": "synthetic",
            "config": {"workload": f"BASELINE config 5 shape: dsp_sipm, {n}{' per GPU' if world > 1 else ''} x {L} f32 traces", "traces_per_gpu": n, "samples": L,
                       "dt_ns": 16.0, "dsp_config": "reference test/test_dsp_sipm.jl:38-68 + sg.wl = 200 ns",
                       "ragged_elements_per_trace": elems / n,
                       "gather": "rccl: scalar table + per trigger group counts -> scan on root -> payload (per-peer sizes)" if world > 1 else "none"},
            "roofline": roofline(achieved, ctx.last_kernel_name(), kms, bytes_per_trace, measured_traffic(ctx.last_kernel_name(), n, L)),
        }
    del wf, bufs
    return res


# ------------------------------------------------------------------------------------------------------------------

def run_rank(args):
    sys.path.insert(0, ROOT)
    env = Env(args)
    multi = env.world > 1
    try:
        if args.workload == "sipm":
            n = args.n if args.n is not None else 625_000
            L = args.L if args.L is not None else 16384
            res = bench_sipm(env, args, n, L)
        else:
            n = args.n if args.n is not None else (1_250_000 if (multi and args.workload == "icpc") else 1_000_000)
            L = args.L if args.L is not None else 8192
            res, wf = bench_icpc(env, args, n, L, args.workload)
            if (not multi and not env.dry and args.workload == "icpc" and not args.no_secondary and env.rank == 0
                    and args.n is None and args.L is None):
                sec = []
                r2, _ = bench_icpc(env, args, n, L, "pz_trap", wf=wf)      # BASELINE config 2 on the same batch
                sec.append(r2)
                del wf
                env.torch.cuda.empty_cache()
                sec.append(bench_sipm(env, args, 625_000, 16384))         # BASELINE config 5's single-GPU shard
                # parameter sets beside the headline's: a trace shorter than the tile -> the bounded instantiation; a length that is no
                # multiple of four samples -> the same instantiation since round 4 (4-byte aligned rows; its batch size differs from the
                # line before so that the profiles can tell the two apart); a Savitzky-Golay window of 27 taps -> icpc_kernel
                # (the generic kernel: what leaving the lean kernel costs); CUSP and ZAC optimised separately -> the two-pass instantiation
                import legenddsp_jl_amd as ldsp
                us = ldsp.us
                r_short, _ = bench_icpc(env, args, 262_144, 8000, "icpc", label="262144 x 8000 f32: a trace shorter than the tile (8192) -> the bounded instantiation of the same kernel")
                sec.append(r_short)
                r_odd, _ = bench_icpc(env, args, 196_608, 8190, "icpc", label="196608 x 8190 f32: a length that is no multiple of four samples (4-byte aligned rows) -> the bounded instantiation of the same kernel (the generic kernel until round 3)")
                sec.append(r_odd)
                r_gen, _ = bench_icpc(env, args, 262_144, 8192, "icpc", pars_filter={"sg": {"wl": 430.0 * ldsp.ns}},
                                      label="fallback: 262144 x 8192 f32 with a Savitzky-Golay window of 430 ns (27 taps; the lean kernel takes 25 = the end of the reference's scan grid) -> generic icpc_kernel")
                sec.append(r_gen)
                r_sep, _ = bench_icpc(env, args, 262_144, 8192, "icpc", pars_filter={"cusp": {"rt": 4.0 * us, "ft": 1.5 * us}, "zac": {"rt": 5.5 * us, "ft": 2.0 * us}},
                                      label="262144 x 8192 f32, CUSP and ZAC optimised separately (pars_filter): two passes of the closed-form stage in one launch")
                sec.append(r_sep)
                res["secondary"] = sec
        if env.rank == 0:
            print(json.dumps(res), flush=True)
    finally:
        env.close()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
