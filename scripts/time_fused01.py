#!/usr/bin/env python3
"""EXPERIMENT: main_trace + LDE against the fused variant whose first inverse NTT pass generates blocks 0-1 of the main trace
from the 372-B trace (zkir_commit_fused01_launch).  Same LDE matrix required; HIP-event medians, the two variants alternating."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zkir_amd import pipeline as pl, runtime as rt, spec, stark

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
log = rt.interpret(spec.fib_endless_program().to_bytes(), [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
ctx = stark.StarkContext(k)
W = stark.W_MAIN
lib = rt.lib()
lib.zkir_commit_fused01_launch.restype = C.c_int
lib.zkir_commit_fused01_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
m = torch.empty((W // 8, n, 8), dtype=torch.int32, device="cuda")
L0 = torch.empty((W // 8, 2 * n, 8), dtype=torch.int32, device="cuda")
L1 = torch.empty_like(L0)


def plain():
    pl._check(lib.zkir_main_trace_launch(C.byref(tr.c), n, 0, m.data_ptr(), sp()))
    pl._check(lib.zkir_lde_launch(ctx.handle, m.data_ptr(), W, L0.data_ptr(), sp()))


def fused():
    pl._check(lib.zkir_commit_fused01_launch(ctx.handle, C.byref(tr.c), n, m.data_ptr(), W, L1.data_ptr(), sp()))


plain(); fused(); torch.cuda.synchronize()
assert torch.equal(L0, L1), "fused01: LDE differs"
for rep in range(3):
    for name, f in (("main_trace + lde", plain), ("fused blocks 0-1  ", fused)):
        ts = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print(f"2^{k} x {W}  {name}: median {np.median(ts):.4f} ms  min {min(ts):.4f}  (pass {rep + 1})")
