#!/usr/bin/env python3
"""Mode-4 proof of the SHA-256 hash chain (BASELINE configs[4]'s program) at 2^k cycles: host witness, prove (best of 3, stage split), both costs of the hash tape
(ZKIR_PROVE_TIMES=1 prints the prover's host phases), the host verifier."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkir_amd import pipeline as pl, runtime as rt, spec, stark
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
blob = spec.sha256_chain_program().to_bytes()
log = rt.interpret(blob, [], rt.VMConfig(max_cycles=1 << k, enable_execution_trace=True))
ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr)); torch.cuda.synchronize()
t0 = time.perf_counter(); pub = rt.public_inputs(log, blob, [], wide_mode=True, mem_witness="host"); t_wit = (time.perf_counter() - t0) * 1e3
ctx = stark.StarkContext(k)
best = None
for _ in range(3):
    t0 = time.perf_counter(); proof, st = stark.prove(ctx, tr, pub, want_stage_ms=True); dt = (time.perf_counter() - t0) * 1e3
    best = dt if best is None else min(best, dt)
t0 = time.perf_counter(); rc = rt.verify(proof, pub); t_ver = (time.perf_counter() - t0) * 1e3
print(f"sha chain 2^{k}: {pub._mem_ref.n_hash_calls} hash calls, host witness {t_wit:.1f} ms, prove {best:.1f} ms (device stages {sum(st):.1f} ms), proof {len(proof) * 4 / 1e6:.1f} MB, verify {t_ver:.1f} ms -> {rc}")
