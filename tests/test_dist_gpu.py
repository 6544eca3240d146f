"""The collectives of the N > 1 path on the hardware's own library: a one-rank `nccl` (= RCCL) group on the box's GPU takes the calls
`legenddsp_jl_amd.dist` issues — the [n, 48] table gathered into row blocks of the result, blocking and `async_op` under the next
batch's kernel, the MIN all-reduce of the argument check, the int64 gather of the counts — and `bench.py` runs under the driver's
launcher (`python -m torch.distributed.run`).  Two ranks need two GPUs: the data movement itself is covered by the world-size-2 / 4 / 8
gloo tests (tests/test_dist_cpu.py).  Each check runs in a child process (a process group per process; this one keeps its GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_rccl_one_rank_group_takes_the_gathers():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1_check.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl world-1 check ok" in r.stdout


def test_bench_under_the_drivers_launcher_one_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",   # (no --n: the launcher's own parser claims the abbreviation)
           "--cpu-sample", "0", "--no-secondary"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["value"] > 1e6 and rec["roofline"]["kernel"] == "lean3::icpc_lean3_kernel"
