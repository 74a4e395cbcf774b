"""The N > 1 code path of bench.py (captured forward+backward, eager gradient exchange + optimiser, rank-0-only
roofline pass) run as two ranks on ONE GPU with the gloo backend: a functional check that it neither hangs nor
diverges in structure -- never a measurement.  (RCCL refuses two ranks on one device; the collective itself is
covered by the 2-rank gloo test of ddp.GradReducer on the CPU.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_two_ranks_one_gpu_gloo():
    env = dict(os.environ, M1_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "C1", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=540)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["scaling"] == "weak"
    assert d["config"]["graph_error"] is None and d["value"] > 0
    assert d["roofline"] is not None and d["cpu_baseline"] is None
