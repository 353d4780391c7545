#!/usr/bin/env python3
"""End-to-end prove timing (zkir_prove stage breakdown) on a 2^k-cycle fib trace.  (The proofs are checked against the oracle in tests/test_gpu_stark.py.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkir_amd import pipeline as pl, runtime as rt, spec, stark
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
t0 = time.perf_counter()
log = rt.interpret(spec.fib_endless_program().to_bytes(), [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
t_host = time.perf_counter() - t0
ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr)); torch.cuda.synchronize()
ctx = stark.StarkContext(k)
pub = rt.public_inputs(log, spec.fib_endless_program().to_bytes())
names = ["main_trace", "lde", "trace_merkle", "lookup_aux", "quotient+merkle", "openings", "deep", "fri", "queries"]
best = None
for it in range(3):
    t0 = time.perf_counter(); proof, ms = stark.prove(ctx, tr, pub, want_stage_ms=True); wall = (time.perf_counter() - t0) * 1e3
    if best is None or wall < best[0]: best = (wall, ms)
wall, ms = best
for nm, v in zip(names, ms): print(f"{nm:16s} {v:9.3f} ms")
print(f"prove wall       {wall:9.3f} ms   (sum of stages {sum(ms):.3f});  host interpret {t_host*1e3:.1f} ms;  proof {len(proof)*4/1024:.1f} KiB")
