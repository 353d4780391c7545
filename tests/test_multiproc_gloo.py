"""N>1 path on CPU: two (and four) gloo ranks shard the rows exactly as bench.py does on GPUs — rank g executes rows [0, g*n/2) of the run
UNTRACED and traces its own rows [g*n/2, (g+1)*n/2) (zkir_interpret_window: own register snapshot, no transport between the ranks,
no data-path collective), cuts its commit shard and its overlapping segment shard out of the window, expands them with the numpy
stand-in for K1, and an all_gather of per-shard digests reproduces the trace of the whole run."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zkir_amd import runtime as rt, spec

import helpers

N_ROWS = 8 * 256


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, blob, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = N_ROWS // world
    lo = rank * (per - 1)                                   # the window covers the commit shard and the segment-proof shard (one row earlier per rank)
    win = rt.interpret(blob, config=rt.VMConfig(max_cycles=N_ROWS, enable_execution_trace=True), tile_rows=256, window=(lo, (rank + 1) * per))
    assert win.cycle_base == lo and win.n_rows == (rank + 1) * per - lo and win.window_open == (rank < world - 1)
    sh = win.shard(rank * per, (rank + 1) * per)
    seg = win.shard(rank * (per - 1), rank * (per - 1) + per)
    rows = helpers.expand_delta_log(sh)
    rows["cycle"] += np.uint64(sh.cycle_base)
    seg_rows = helpers.expand_delta_log(seg)
    seg_rows["cycle"] += np.uint64(seg.cycle_base)
    digest = torch.tensor(list(hashlib.sha256(rows.tobytes()).digest()), dtype=torch.uint8)
    seg_digest = hashlib.sha256(seg_rows.tobytes()).digest()
    all_seg = [None] * world
    dist.all_gather_object(all_seg, seg_digest)
    gathered = [torch.zeros(32, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gathered, digest)                       # the only collective of the path: 32 bytes per rank
    n_ev = torch.tensor([len(sh.reg_events)], dtype=torch.int64)
    dist.all_reduce(n_ev)
    if rank == 0:
        out_q.put(([bytes(g.tolist()) for g in gathered], int(n_ev.item()), all_seg))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharding_gloo(world):
    """world = 2 and 4 (VERDICT r4 task 7): every rank's commit shard and overlapping segment shard, gathered, are the rows of the one-process run."""
    blob = spec.sha256_chain_program().to_bytes()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, blob, q)) for r in range(world)]
    for p in procs:
        p.start()
    digests, n_ev, seg_digests = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    log = rt.interpret(blob, config=rt.VMConfig(max_cycles=N_ROWS, enable_execution_trace=True), tile_rows=256)
    full = helpers.expand_delta_log(log)
    per = N_ROWS // world
    assert digests == [hashlib.sha256(full[g * per:(g + 1) * per].tobytes()).digest() for g in range(world)]
    assert n_ev >= len(log.reg_events)          # each shard carries its own 16 snapshot events
    assert seg_digests == [hashlib.sha256(full[g * (per - 1):g * (per - 1) + per].tobytes()).digest() for g in range(world)]
