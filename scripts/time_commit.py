#!/usr/bin/env python3
"""Per-stage timing of the commit pipeline (trace_fill -> main_trace -> LDE -> Merkle) with HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkir_amd import pipeline as pl, runtime as rt, spec, stark

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
log = rt.interpret(spec.fib_endless_program().to_bytes(), [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); fa = pl.trace_fill_args(ddl, tr)
ctx = stark.StarkContext(k)
W = int(rt.lib().zkir_main_trace_width())          # the committed width of the library that is loaded (variants differ)
m = torch.empty((W // 8, n, 8), dtype=torch.int32, device="cuda")
L = torch.empty((W // 8, 2 * n, 8), dtype=torch.int32, device="cuda")
tree = torch.empty(4 * (4 * n - 1), dtype=torch.int32, device="cuda")
import ctypes as C
lib = rt.lib(); sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def stages():
    yield "trace_fill", lambda: pl.trace_fill(fa)
    yield "main_trace", lambda: pl._check(lib.zkir_main_trace_launch(C.byref(tr.c), n, 0, m.data_ptr(), sp()))
    yield "lde", lambda: pl._check(lib.zkir_lde_launch(ctx.handle, m.data_ptr(), W, L.data_ptr(), sp()))
    yield "merkle", lambda: pl._check(lib.zkir_merkle_commit_launch(ctx.handle, L.data_ptr(), W, 2 * n, tree.data_ptr(), sp()))
for _ in range(3):
    for name, f in stages(): f()
torch.cuda.synchronize()
tot = 0
for name, f in stages():
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{name:12s} {np.median(ts):9.3f} ms   (min {min(ts):.3f})")
    tot += np.median(ts)
print(f"total        {tot:9.3f} ms  -> {n / tot / 1e3:.1f} M rows/s   root={tree[-4:].cpu().numpy().view(np.uint32)}")
