"""bench.py's N > 1 path executed for real: two ranks launched with torch.distributed.run exactly as the driver does
(`--nproc-per-node 2 ... bench.py --gpus 2`).  With two GPUs visible the ranks use RCCL (backend nccl, one device each); on a
one-GPU box both ranks share cuda:0 and the 16-byte root exchange goes through gloo (ZKIR_BENCH_SHARE_GPU=1) — the sharding
(every rank executes the rows before its shard untraced and traces its own: zkir_interpret_window / zkir_exec_window), the per-shard
commitment, the all-gather and the Merkle cap are the same code either way.
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
    lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 12_000, "bench.py prints ONE small JSON line (the driver parses it)"
    head = json.loads(lines[0])
    detail = json.loads([x for x in p.stderr.splitlines() if x.startswith("bench_detail: ")][-1][len("bench_detail: "):])     # the full record goes to stderr / bench_detail.json
    for key in ("value", "n_gpus", "ms_per_step", "merkle_root"):
        assert head[key] == detail[key], key
    assert head.get("roofline") and head["roofline"].get("frac") is not None
    return detail


def test_world1_over_rccl():
    """The nccl (= RCCL) process group for real, on a box with ONE GPU: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with
    ZKIR_BENCH_FORCE_DIST=1 takes bench.py's whole N > 1 code path at world size 1 — init_process_group("nccl", device_id=..), the window
    interpretation, the DEVICE all-gather of the int32 root, the cap, the f64 all-reduces, the barriers, all_gather_object /
    broadcast_object_list / gather_object, the segment proofs and zkir_verify_chain.  The capped root of one shard is the shard's own root,
    and it must equal the oracle's commitment of the run."""
    k = 14
    out = _launch(1, k, {"ZKIR_BENCH_FORCE_DIST": "1"})
    n = 1 << k
    pg = out["process_group"]
    assert pg and pg["backend"] == "nccl" and pg["world_size"] == 1 and pg["forced_at_world_1"]
    assert out["n_gpus"] == 1 and out["config"]["rows_per_gpu"] == n and out["allgather_cap_ms"] is not None
    rows = oracle.run(spec.fib_endless_program().to_bytes(), max_cycles=n, enable_execution_trace=True).rows
    want = list(map(int, so.commit_trace(rows, 1)))
    assert out["merkle_roots_all_ranks"] == [want] and out["merkle_root"] == want
    e2e = out["multi_gpu_end_to_end"]
    assert e2e and "error" not in e2e, e2e
    assert e2e["one_run_row_sharded"]["root"] == want
    sp = out["segment_prove"]
    assert sp and "error" not in sp, sp
    assert sp["segments"] == 2 and sp["verify_chain_code"] == 0                     # rows [0, n) and the one-row tail [n - 1, n)
    assert 0.5 < out["efficiency_vs_same_size_single_gpu"] <= 1.05


def test_two_ranks_sharded_commitment():
    import torch
    k, world = 14, 2
    two_gpus = torch.cuda.device_count() >= 2
    out = _launch(world, k, {} if two_gpus else {"ZKIR_BENCH_BACKEND": "gloo", "ZKIR_BENCH_SHARE_GPU": "1"})
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["rows_per_gpu"] == 1 << k
    # no shard transport: every rank executed its own prefix of the run (rank g: g n rows untraced, n rows traced)
    assert out["host_interpretations_per_node"] == 2 and out["shard_distribution_s"] == 0.0 and len(out["host_window_s_per_rank"]) == 2
    assert out["value"] > 0 and out["allgather_cap_ms"] is not None
    n = 1 << k
    rows = oracle.run(spec.fib_endless_program().to_bytes(), max_cycles=world * n, enable_execution_trace=True).rows
    want = [so.commit_trace(rows[g * n:(g + 1) * n], 1) for g in range(world)]
    assert [list(map(int, r)) for r in want] == out["merkle_roots_all_ranks"]
    assert list(map(int, so.compress(want[0], want[1]))) == out["merkle_root"]
    # the hand-off with the HOST IN THE LOOP: zkir_exec_window on every rank (fast-forward + streamed traced window) -> commit -> all-gather + cap
    # gives the same capped root as the resident-shard steps, and the bench line carries the end-to-end figures of both modes
    e2e = out["multi_gpu_end_to_end"]
    assert e2e and "error" not in e2e, e2e
    assert e2e["one_run_row_sharded"]["root"] == out["merkle_root"] and e2e["one_run_row_sharded"]["rows"] == world * n
    assert out["end_to_end_rows_per_s_incl_host"] == e2e["one_run_row_sharded"]["end_to_end_rows_per_s_incl_host"] > 0
    assert e2e["independent_runs_one_per_gpu"]["runs"] == world and e2e["independent_runs_one_per_gpu"]["rows_per_s_end_to_end_incl_host"] > 0
    # the run PROVEN in segments, one per rank (+ the one-row tail), and accepted as one run by zkir_verify_chain on rank 0
    sp = out["segment_prove"]
    assert sp and "error" not in sp, sp
    assert sp["segments"] == world + 1 and sp["verify_chain_code"] == 0 and sp["rows_per_segment"] == n and sp["ms_all_segments_in_parallel"] > 0


def test_non_power_of_two_world_is_rejected():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "power of two" in (p.stdout + p.stderr)


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2", reason="the nccl (RCCL) branch needs two GPUs; on a one-GPU box the same path runs over gloo above")
def test_two_ranks_over_rccl():
    """bench.py --gpus 2 with backend nccl, one device per rank: init_process_group("nccl"), the device all-gather of the 16-byte roots,
    the all-reduces of the timings.  (Until a box with two GPUs runs this, the nccl branch has never executed: DESIGN.md §4 says so.)"""
    out = _launch(2, 14, {})
    n = 1 << 14
    rows = oracle.run(spec.fib_endless_program().to_bytes(), max_cycles=2 * n, enable_execution_trace=True).rows
    want = [so.commit_trace(rows[g * n:(g + 1) * n], 1) for g in range(2)]
    assert list(map(int, so.compress(want[0], want[1]))) == out["merkle_root"]
    assert out["multi_gpu_end_to_end"]["one_run_row_sharded"]["root"] == out["merkle_root"]
