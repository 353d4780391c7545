"""Pipelined proving of independent runs on one GPU: host threads interpret (the VM is sequential — one run per thread) and
upload the delta log on their own HIP streams while the single GPU thread fills the trace and proves the previous run.

This is the caller side of the hot path (SURVEY §8f widening): what a proving service built on the C ABI looks like — one
process per GPU, a few host cores feeding it, the execution trace never leaving HBM.  A `zkir_stark_ctx` serves one proof at a time (its
workspace), so there is one consumer per context; the producers overlap with it because the C calls and the copies release the GIL.
"""
from __future__ import annotations

import queue
import threading
import time
from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import pipeline as pl, runtime as rt, stark

Job = Tuple[bytes, Sequence[int], rt.VMConfig]
_spare_ctx: dict = {}                                         # log2_rows -> idle StarkContexts of the extra proving threads
_spare_lock = threading.Lock()
_SPARE_MAX = 2                                                # contexts kept between calls (each holds tables + a proof workspace of GBs at large sizes)


def close_spare_contexts() -> None:
    """Release the contexts the extra proving threads keep between prove_many calls (device tables + workspace)."""
    with _spare_lock:
        for lst in _spare_ctx.values():
            for c in lst:
                c.close()
        _spare_ctx.clear()


def _park_ctx(log2_rows: int, c) -> None:
    with _spare_lock:
        if sum(len(v) for v in _spare_ctx.values()) >= _SPARE_MAX:      # cap: drop the oldest parked context of another size first
            for k in list(_spare_ctx):
                if k != log2_rows and _spare_ctx[k]:
                    _spare_ctx[k].pop(0).close()
                    break
            else:
                c.close()
                return
        _spare_ctx.setdefault(log2_rows, []).append(c)


@dataclass
class PipelineReport:
    runs: int = 0
    rows: int = 0
    wall_s: float = 0.0
    interpret_s: float = 0.0          # summed over runs (overlapped across producer threads)
    upload_s: float = 0.0
    proofs: List[np.ndarray] = field(default_factory=list)

    @property
    def ms_per_run(self) -> float:
        return self.wall_s / max(self.runs, 1) * 1e3

    @property
    def rows_per_s(self) -> float:
        return self.rows / self.wall_s if self.wall_s else 0.0


def prove_many(jobs: Iterable[Job], log2_rows: int, producers: int = 3, ctx: Optional[stark.StarkContext] = None,
               keep_proofs: bool = True, commit_only: bool = False, provers: int = 2) -> PipelineReport:
    """Prove every job (program blob, inputs, VMConfig with enable_execution_trace; the run's row count must pad to 2^log2_rows,
    i.e. 2^(log2_rows-1) < rows <= 2^log2_rows).  Proofs come back in job order.  commit_only: stop after the trace commitment
    (trace fill + main trace + LDE + Merkle); `proofs` then holds the 4-word roots.
    provers: a proof makes ~13 host round trips (Fiat-Shamir on the host) during which ITS stream is idle; with two proving threads,
    each with its own context (workspace) and stream, the GPU works on one proof while the other waits for its transcript."""
    pl._require_gpu()
    jobs = list(jobs)
    own_ctx = ctx is None
    ctx = ctx or stark.StarkContext(log2_rows)
    ready: "queue.Queue" = queue.Queue(maxsize=max(2, producers))
    todo = list(enumerate(jobs))[::-1]
    lock = threading.Lock()
    errors: List[BaseException] = []
    stop = threading.Event()                                  # set on any failure (main thread included): every helper thread leaves

    def take():
        """Next ready item, or None once the pipeline is stopping (never blocks forever: a failed consumer stops feeding the queue)."""
        while not stop.is_set():
            try:
                return ready.get(timeout=0.05)
            except queue.Empty:
                continue
        return None

    def producer():
        s = torch.cuda.Stream()
        while True:
            with lock:
                if not todo or errors or stop.is_set():
                    return
                idx, (blob, inputs, cfg) = todo.pop()
            try:
                t0 = time.perf_counter()
                log = rt.interpret(blob, list(inputs), cfg)
                t1 = time.perf_counter()
                if log.n_rows == 0 or stark.padded_log_n(log.n_rows) != log2_rows:
                    raise rt.RuntimeError(rt.ERR_ARGUMENT, f"job {idx}: {log.n_rows} trace rows do not pad to the context's 2^{log2_rows}")
                pub = rt.public_inputs(log, blob, list(inputs), cfg.enable_deferred_model)
                with torch.cuda.stream(s):
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record(s)
                    ddl = pl.upload(log)
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(s)
                ev.synchronize()
                log.close()                                   # (the copies are done: the log's pinned blocks go back to the pool)
                ready.put((idx, ddl, ev, t1 - t0, ev0.elapsed_time(ev) * 1e-3, pub, log.n_rows))      # upload_s: the copies' time on the copy stream (HIP events), not the wait for a queue slot
            except BaseException as e:                    # surfaced by the consumer
                errors.append(e)
                stop.set()
                return

    rep = PipelineReport(runs=len(jobs), rows=0)
    out: List[Optional[np.ndarray]] = [None] * len(jobs)
    threads = [threading.Thread(target=producer, daemon=True) for _ in range(max(1, producers))]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    pending = None                                        # commit_only: the previous run's (index, pinned root, event, its buffers)
    roots = [torch.empty(4, dtype=torch.int32, pin_memory=True) for _ in range(2)] if commit_only else []
    commit_streams = [torch.cuda.Stream() for _ in range(2)] if commit_only else []
    idx_run = 0

    def finish(p):
        if p is not None:
            p[2].synchronize()
            if keep_proofs:
                out[p[0]] = p[1].numpy().copy().view(np.uint32)

    # ---- extra proving threads (full proofs only): each takes whole jobs off the queue, on its own context and stream ----
    n_extra = 0 if commit_only else max(0, min(provers, len(jobs)) - 1)
    claimed = [0]

    def claim() -> bool:
        with lock:
            if claimed[0] >= len(jobs) or errors or stop.is_set():
                return False
            claimed[0] += 1
            return True

    def prover_thread():
        try:
            with _spare_lock:                                 # contexts of the extra threads are kept between calls (tables + workspace)
                my_ctx = _spare_ctx[log2_rows].pop() if _spare_ctx.get(log2_rows) else None
            my_ctx = my_ctx or stark.StarkContext(log2_rows)
            stream = torch.cuda.Stream()
            try:
                with torch.cuda.stream(stream):
                    while claim():
                        item = take()
                        if item is None:
                            return
                        idx, ddl, ev, h, u, pub, n_rows = item
                        stream.wait_event(ev)
                        tr = pl.DeviceTrace(ddl)
                        pl.trace_fill(pl.trace_fill_args(ddl, tr))
                        proof = stark.prove(my_ctx, tr, pub)
                        with lock:
                            rep.interpret_s += h; rep.upload_s += u; rep.rows += n_rows
                            if keep_proofs:
                                out[idx] = proof
            finally:
                _park_ctx(log2_rows, my_ctx)
        except BaseException as e:                            # noqa: BLE001
            with lock:
                errors.append(e)
            stop.set()

    extra = [threading.Thread(target=prover_thread, daemon=True) for _ in range(n_extra)]
    for t in extra:
        t.start()
    try:
        while claim():
            item = take()
            if item is None:
                break
            idx, ddl, ev, h, u, pub, n_rows = item
            with lock:
                rep.interpret_s += h
                rep.upload_s += u
                rep.rows += n_rows
            if commit_only:
                # The commitment needs nothing from the host: queue this run's kernels BEHIND the previous run's, and only then wait
                # for the previous root (16 bytes into pinned memory) — the GPU goes from one run to the next without a gap.  Runs
                # alternate between two streams, so the latency-bound tail of one tree (a few workgroups walking the top levels) shares
                # the device with the bandwidth-bound head of the next run.
                cs = commit_streams[idx_run & 1]
                with torch.cuda.stream(cs):
                    cs.wait_event(ev)
                    tr = pl.DeviceTrace(ddl)
                    pl.trace_fill(pl.trace_fill_args(ddl, tr))
                    m = stark.main_trace(tr, deferred=bool(pub.deferred))
                    L = stark.lde(ctx, m, clobber=True)
                    tree = stark.merkle_commit(ctx, L, stark.W_MAIN)
                    root = roots[idx_run & 1]             # two pinned landing buffers, used alternately (pinned allocation is slow)
                    idx_run += 1
                    root.copy_(tree[-4:], non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(cs)
                finish(pending)
                pending = (idx, root, done, (ddl, tr, m, L, tree))
                continue
            torch.cuda.current_stream().wait_event(ev)
            tr = pl.DeviceTrace(ddl)
            pl.trace_fill(pl.trace_fill_args(ddl, tr))
            proof = stark.prove(ctx, tr, pub)             # returns after the proof words are on the host: the run's buffers are idle
            if keep_proofs:
                out[idx] = proof
        finish(pending)
        pending = None
        for t in extra:
            t.join()
        if errors:
            raise errors[0]
        torch.cuda.synchronize()
        rep.wall_s = time.perf_counter() - t0
    except BaseException as e:                            # a failure on THIS thread (e.g. stark.prove) stops the helpers too
        with lock:
            errors.append(e)
        raise
    finally:
        stop.set()
        with lock:
            todo.clear()
        while any(t.is_alive() for t in threads):         # unblock producers stuck on a full queue
            try:
                ready.get_nowait()
            except queue.Empty:
                time.sleep(0.001)
        for t in extra:                                   # the proving threads poll `stop`: they hold a context and a stream until they leave
            t.join()
        if own_ctx:
            ctx.close()
    rep.proofs = [p for p in out if p is not None]
    return rep
