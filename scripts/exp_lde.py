#!/usr/bin/env python3
"""LDE alone at 2^k rows x W columns (random canonical values), HIP-event timing; for pass / tile experiments (scripts/exp_lde.sh adds the per-kernel split)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkir_amd import runtime as rt, stark
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 152
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = 1 << k
ctx = stark.StarkContext(k)
m0 = torch.randint(0, 2013265921, (W // 8, n, 8), dtype=torch.int32, device="cuda")
m = torch.empty_like(m0); L = torch.empty((W // 8, 2 * n, 8), dtype=torch.int32, device="cuda")
lib = rt.lib(); sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
ts = []
for i in range(reps + 3):
    m.copy_(m0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); rc = lib.zkir_lde_launch(ctx.handle, m.data_ptr(), W, L.data_ptr(), sp()); b.record(); torch.cuda.synchronize()
    assert rc == 0
    if i >= 3: ts.append(a.elapsed_time(b))
print(f"lde 2^{k} x {W}: median {np.median(ts):.3f} ms, min {min(ts):.3f}; per column of 2^20 rows {np.median(ts) / W / (n / 2**20) * 1e3:.2f} us")
