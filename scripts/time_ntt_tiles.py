#!/usr/bin/env python3
"""EXPERIMENT: one strided NTT pass over 152 columns with different tile geometries, side by side — does a pass run faster per stage when two (or four)
independent workgroups share a CU instead of one lock-stepped 128 KiB tile?  HIP-event medians; bytes per pass = 8 B per element whatever the geometry."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from zkir_amd import pipeline as pl, runtime as rt, stark

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = stark.W_MAIN
lib = rt.lib()
lib.zkir_ntt_strided_variant_launch.restype = C.c_int
lib.zkir_ntt_strided_variant_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
ctx = stark.StarkContext(k)
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
names = {0: "10 stages, 1024 x 4 pos (128 B rows), 128 KiB, 1 wg/CU [lde_run]", 1: "10 stages, 1024 x 2 pos ( 64 B rows),  64 KiB, 2 wg/CU",
         2: " 8 stages,  256 x 8 pos (256 B rows),  64 KiB, 2 wg/CU", 3: " 8 stages,  256 x 4 pos (128 B rows),  32 KiB, 4 wg/CU", 4: " 6 stages,   64 x 16 pos (512 B rows),  32 KiB, 4 wg/CU"}
stages = {0: 10, 1: 10, 2: 8, 3: 8, 4: 6}
for fwd in (0, 1):
    n = (2 << k) if fwd else (1 << k)
    data = torch.randint(0, 2013265921, (W // 8, n, 8), dtype=torch.int32, device="cuda")
    gb = 8 * W * n / 1e9
    for rep in range(2):
        for v in range(5):
            f = lambda: pl._check(lib.zkir_ntt_strided_variant_launch(ctx.handle, data.data_ptr(), W, v, fwd, sp()))  # noqa: E731
            f(); torch.cuda.synchronize()
            ts = []
            for _ in range(15):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            ms = float(np.median(ts))
            print(f"{'forward 2^' + str(k + 1) if fwd else 'inverse 2^' + str(k)} x {W}  {names[v]}: {ms:.4f} ms = {gb / ms:.2f} TB/s, {ms / stages[v] * 1e3:.1f} us per stage  (pass {rep + 1})")
    del data
