#!/usr/bin/env python3
"""the leaf hash overlapped with the LDE (zkir_commit_overlapped_launch: block-group-wise sponge absorption on a second stream) against the two stages one
after the other, on the REAL main trace of the 2^k-cycle fib run (152 columns).  Prints ms per commitment (HIP events, median of `reps`) and checks that every variant's tree
equals the baseline's.  Usage: time_overlap.py [k=20] [reps=20] [groups=1,2,4,8]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkir_amd import pipeline as pl, runtime as rt, spec, stark
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
groups = [int(g) for g in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]
n = 1 << k
lib = rt.lib()
lib.zkir_commit_overlapped_launch.restype = C.c_int
lib.zkir_commit_overlapped_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
blob = spec.fib_endless_program().to_bytes()
log = rt.interpret(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
ctx = stark.StarkContext(k)
m0 = stark.main_trace(tr)
W = stark.W_MAIN
m = torch.empty_like(m0); L = torch.empty((W // 8, 2 * n, 8), dtype=torch.int32, device="cuda"); tree = torch.empty(4 * (4 * n - 1), dtype=torch.int32, device="cuda")
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(fn):
    ts = []
    for i in range(reps + 3):
        m.copy_(m0); tree.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if i >= 3:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(min(ts)), tree.clone()


def base():
    assert lib.zkir_lde_launch(ctx.handle, m.data_ptr(), W, L.data_ptr(), sp()) == 0
    assert lib.zkir_merkle_commit_launch(ctx.handle, L.data_ptr(), W, 2 * n, tree.data_ptr(), sp()) == 0


med0, min0, want = run(base)
print(f"2^{k} rows x {W} columns, {reps} reps: LDE then Merkle (one stream): median {med0:.3f} ms, min {min0:.3f}", flush=True)
for g in groups:
    def ov(g=g):
        assert lib.zkir_commit_overlapped_launch(ctx.handle, m.data_ptr(), W, L.data_ptr(), tree.data_ptr(), g, sp()) == 0
    med, mn, got = run(ov)
    ok = torch.equal(got, want)
    print(f"  overlapped, groups of {g} block(s) ({-(-W // 8 // g)} groups): median {med:.3f} ms, min {mn:.3f}  ({(med / med0 - 1) * 100:+.1f} %)  tree == baseline: {ok}", flush=True)
    assert ok
med1, min1, _ = run(base)
print(f"LDE then Merkle again: median {med1:.3f} ms, min {min1:.3f}")
