#!/usr/bin/env python3
"""HIP-event timing of the commit stages at 2^k rows (default 20), for kernel tuning: main_trace, lde, merkle."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkir_amd import pipeline as pl, runtime as rt, spec, stark
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
log = rt.interpret(spec.fib_endless_program().to_bytes(), [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
ctx = stark.StarkContext(k); W = stark.W_MAIN
m = torch.empty((W // 8, n, 8), dtype=torch.int32, device="cuda"); L = torch.empty((W // 8, 2 * n, 8), dtype=torch.int32, device="cuda")
tree = torch.empty(4 * (4 * n - 1), dtype=torch.int32, device="cuda")
lib = rt.lib(); sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
st = [("main_trace", lambda: lib.zkir_main_trace_launch(C.byref(tr.c), n, 0, m.data_ptr(), sp())),
      ("lde", lambda: lib.zkir_lde_launch(ctx.handle, m.data_ptr(), W, L.data_ptr(), sp())),
      ("merkle", lambda: lib.zkir_merkle_commit_launch(ctx.handle, L.data_ptr(), W, 2 * n, tree.data_ptr(), sp()))]
for _ in range(5):
    for _, f in st: f()
torch.cuda.synchronize()
for name, f in st:
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{name:12s} {np.median(ts):8.3f} ms (min {min(ts):.3f})")
