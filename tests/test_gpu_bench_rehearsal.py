"""bench.py's one-process-per-GPU path, rehearsed on the one-GPU box: two ranks pinned to device 0, gloo in place of RCCL
(which refuses two ranks on one device).  Everything else is the code the driver runs with --gpus N: row-split weights per
rank, the concat after every MUL_MAT group, the barrier + max-over-ranks timing, the roofline leg on each rank, rank 0's
JSON line.  The throughput it prints is meaningless (gloo stages through the host); the test checks that the path runs
and that the line has the contract's fields."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]


def test_bench_two_ranks_on_one_gpu_with_gloo():
    env = dict(os.environ, QMM_BENCH_DEVICE="0", QMM_BENCH_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--n-gen", "4"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout + p.stderr)[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 1 and d["value"] > 0
    assert d["config"]["parallelism"].startswith("ggml row split over 2 GPUs") and d["config"]["tg_launch"] == "eager"
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm"


def test_bench_split_path_through_rccl_with_a_world_of_one():
    """the N > 1 code of bench.py on real RCCL: a process group of one rank, the exchange forced through the backend, the
    token-generation pass with its all-gathers captured in a hipGraph (QMM_BENCH_FORCE_SPLIT=1)"""
    env = dict(os.environ, QMM_BENCH_FORCE_SPLIT="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--n-gen", "8", "--no-e2e", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout + p.stderr)[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["tg_launch"] == "hipGraph replay", d["config"]
    assert "capture of the split pass failed" not in p.stderr
