#!/usr/bin/env python3
"""H2D of a 2^k-row delta log (pipeline.upload) from the log's own host buffers: pinned pool blocks (default) against pageable memory (ZKIR_PIN_LOG=0).
Per-array copy times (zkir_host_to_device + synchronise) and the whole upload; VERDICT r5 weak #10."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zkir_amd import pipeline as pl, runtime as rt, spec
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
blob = spec.fib_endless_program().to_bytes()
torch.zeros(1 << 20, device="cuda").sum().item()
for rep in range(4):
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=1 << k, enable_execution_trace=True))
    parts = {}
    for name in ("reg_events", "pc", "inst", "tile_snap", "tile_ev_off"):
        a = getattr(log, name)
        flat = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        assert flat.ctypes.data == a.ctypes.data                    # a view of the log's buffer, not a copy
        out = torch.empty(flat.size, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pl._check(rt.lib().zkir_host_to_device(out.data_ptr(), flat.ctypes.data, flat.size, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        parts[name] = (flat.size / 1e6, dt * 1e3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ddl = pl.upload(log)
    torch.cuda.synchronize(); whole = time.perf_counter() - t0
    mb = sum(v[0] for v in parts.values())
    print(f"rep {rep}: upload {whole * 1e3:.2f} ms for {mb:.1f} MB = {mb / whole / 1e3:.1f} GB/s; " + ", ".join(f"{n} {v[0]:.1f} MB {v[1]:.2f} ms ({v[0] / v[1]:.1f} GB/s)" for n, v in parts.items()), flush=True)
    del ddl
    log.close()
