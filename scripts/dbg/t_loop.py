import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
from zkir_amd import pipeline as pl, runtime as rt, spec, stark
n = 1 << 20
blob = spec.memory_loop_program(min(65535, n // 13)).to_bytes()
log = rt.interpret(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr)); torch.cuda.synchronize()
ctx = stark.StarkContext(20)
for mode in (0, 3, 3, 0, 3):
    pub = rt.public_inputs(log, blob, [], mem_mode=mode == 3)
    for it in range(4):
        t0 = time.perf_counter(); proof, ms = stark.prove(ctx, tr, pub, want_stage_ms=True); wall = (time.perf_counter() - t0) * 1e3
        print(mode, it, f"wall {wall:.2f} sum {sum(ms):.2f}", " ".join(f"{v:.2f}" for v in ms))
