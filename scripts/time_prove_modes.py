#!/usr/bin/env python3
"""End-to-end prove timing by proof MODE on spec.memory_ring_program at 2^k rows: mode 0 (default), 2 (+ the I/O argument), 3 (+ the memory argument and the
bitwise opcodes; memory witness on the device), with the zkir_prove stage breakdown, the host's replay of the same memory witness for comparison, and the host verifier's time.
    python scripts/time_prove_modes.py [k=20] [log2_cells=13]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zkir_amd import pipeline as pl, runtime as rt, spec, stark
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lc = int(sys.argv[2]) if len(sys.argv) > 2 else 13
n = 1 << k
blob = spec.memory_ring_program(lc).to_bytes()
t0 = time.perf_counter()
log = rt.interpret(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
t_host = time.perf_counter() - t0
ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr)); torch.cuda.synchronize()
ctx = stark.StarkContext(k)
names = ["main_trace", "lde", "trace_merkle", "lookup_aux", "quotient+merkle", "openings", "deep", "fri", "queries"]
print(f"spec.memory_ring_program({lc}) at 2^{k} rows: host interpret {t_host * 1e3:.1f} ms")
base = None
for mode in (0, 2, 3, 4):
    pub = rt.public_inputs(log, blob, [], io_mode=mode == 2, mem_mode=mode == 3, wide_mode=mode == 4, mem_witness="device")
    best = None
    for it in range(3):
        t0 = time.perf_counter(); proof, ms = stark.prove(ctx, tr, pub, want_stage_ms=True); wall = (time.perf_counter() - t0) * 1e3
        if best is None or wall < best[0]: best = (wall, ms)
    t0 = time.perf_counter(); rc = rt.verify(proof, pub); t_ver = (time.perf_counter() - t0) * 1e3
    assert rc == 0, rc
    base = best[0] if base is None else base
    print(f"mode {mode}: prove {best[0]:9.2f} ms (x {best[0] / base:.2f})  committed {int(proof[3])} + aux  proof {len(proof) * 4 / 1024:.0f} KiB  host verify {t_ver:.1f} ms   " +
          "  ".join(f"{nm} {v:.2f}" for nm, v in zip(names, best[1])))
t0 = time.perf_counter(); w = rt.MemcheckWitness(log, blob); t_w = (time.perf_counter() - t0) * 1e3
print(f"memory witness by the host's sequential replay: {t_w:.1f} ms ({w.n_accesses} accesses, {w.n_cells} cells)")
