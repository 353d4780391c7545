"""bench.py's N > 1 path executed for real: two ranks launched with torch.distributed.run exactly as the driver does
(`--nproc-per-node 2 ... bench.py --gpus 2`).  With two GPUs visible the ranks use RCCL (backend nccl, one device each); on a
one-GPU box both ranks share cuda:0 and the 16-byte root exchange goes through gloo (ZKIR_BENCH_SHARE_GPU=1) — the sharding, the
single interpretation per node, the per-shard commitment, the all-gather and the Merkle cap are the same code either way.
The capped root must equal the oracle's: compress(commit(rows of shard 0), commit(rows of shard 1))."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import api as oracle, stark_api as so
from zkir_amd import spec

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, k, extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--log2-rows", str(k), "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = [x for x in p.stdout.splitlines() if x.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_sharded_commitment():
    import torch
    k, world = 14, 2
    two_gpus = torch.cuda.device_count() >= 2
    out = _launch(world, k, {} if two_gpus else {"ZKIR_BENCH_BACKEND": "gloo", "ZKIR_BENCH_SHARE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["rows_per_gpu"] == 1 << k and out["host_interpretations_per_node"] == 1
    assert out["value"] > 0 and out["allgather_cap_ms"] is not None
    n = 1 << k
    rows = oracle.run(spec.fib_endless_program().to_bytes(), max_cycles=world * n, enable_execution_trace=True).rows
    want = [so.commit_trace(rows[g * n:(g + 1) * n], 1) for g in range(world)]
    assert [list(map(int, r)) for r in want] == out["merkle_roots_all_ranks"]
    assert list(map(int, so.compress(want[0], want[1]))) == out["merkle_root"]
    # the run PROVEN in segments, one per rank (+ the one-row tail), and accepted as one run by zkir_verify_chain on rank 0
    sp = out["segment_prove"]
    assert sp and "error" not in sp, sp
    assert sp["segments"] == world + 1 and sp["verify_chain_code"] == 0 and sp["rows_per_segment"] == n and sp["ms_all_segments_in_parallel"] > 0


def test_non_power_of_two_world_is_rejected():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "power of two" in (p.stdout + p.stderr)
