#!/usr/bin/env python3
"""BASELINE configs[4] shape: SHA-256 hash-chain program, 2^k cycles (default 22) — syscall-chip trace columns on one GPU.
Times (HIP events) and prices against HBM: K1 trace fill, memory-op expansion / CSR / sorted memory trace, SHA-256 chip (K3)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zkir_amd import pipeline as pl, runtime as rt, spec

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << k
t0 = time.perf_counter()
log = rt.interpret(spec.sha256_chain_program().to_bytes(), [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
t_host = time.perf_counter() - t0
n_ops, n_blk = len(log.mem_events), len(log.sha_blocks)
print(f"host interpret: {n} cycles in {t_host:.2f} s ({n / t_host / 1e6:.1f} M cycles/s); {n_ops} memory ops, {n_blk} single-block SHA-256 calls, "
      f"{len(log.reg_events)} register events")
lib = rt.lib()
dev = torch.device("cuda")
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def timeit(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def report(name, ms, nbytes):
    print(f"{name:28s} {ms:8.3f} ms   {nbytes / 1e6:9.1f} MB algorithmic   {nbytes / ms / 1e6:8.1f} GB/s = {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s")


ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); fa = pl.trace_fill_args(ddl, tr)
report("trace_fill (K1)", timeit(lambda: pl.trace_fill(fa)), pl.trace_fill_bytes(ddl))

ev = pl._to_dev(log.mem_events, dev)
rows_c, sort_c = pl.MemopColumns(n_ops, dev), pl.MemopColumns(n_ops, dev)
offs = torch.empty(n + 1, dtype=torch.int64, device=dev); scratch = torch.empty(n, dtype=torch.uint8, device=dev)
report("memops row offsets (CSR)", timeit(lambda: pl._check(lib.zkir_memops_row_offsets_launch(ev.data_ptr(), n_ops, n, offs.data_ptr(), sp()))), 8 * (n + 1))
report("memops expand (row order)", timeit(lambda: pl._check(lib.zkir_memops_expand_launch(ev.data_ptr(), n_ops, 0, C.byref(rows_c.c), sp()))), (24 + 39) * n_ops)
report("memops sort (get_memory_trace)", timeit(lambda: pl._check(lib.zkir_memops_sort_launch(ev.data_ptr(), n_ops, n, 0, offs.data_ptr(), scratch.data_ptr(), C.byref(sort_c.c), sp()))),
       (24 * 2 + 39) * n_ops + n)

blk = pl._to_dev(log.sha_blocks, dev)
out = torch.empty((608, n_blk), dtype=torch.int32, device=dev); ts = torch.empty(n_blk, dtype=torch.int64, device=dev)
report("sha256 chip (K3)", timeit(lambda: pl._check(lib.zkir_sha256_chip_launch(blk.data_ptr(), n_blk, out.data_ptr(), n_blk, ts.data_ptr(), sp()))), (72 + 2432 + 8) * n_blk)
# spot parity: final state of a few blocks equals hashlib
import hashlib
o = out[:, [0, n_blk // 2, n_blk - 1]].cpu().numpy().view(np.uint32)
for col, idx in enumerate([0, n_blk // 2, n_blk - 1]):
    msg = log.sha_blocks[idx]["message_block"].astype(">u4").tobytes()[:32]
    assert o[600:608, col].astype(">u4").tobytes() == hashlib.sha256(msg).digest()
print("sha chip spot-check vs hashlib: ok")
