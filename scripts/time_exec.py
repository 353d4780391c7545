import time, sys, os
sys.path.insert(0, os.getcwd())
import torch
from zkir_amd import runtime as rt, spec
blob = spec.fib_endless_program().to_bytes()
cfg = rt.VMConfig(max_cycles=1 << 20, enable_execution_trace=True)
for i in range(8):
    t0 = time.perf_counter(); res = rt.VM(blob, [], cfg).run(); t1 = time.perf_counter(); st = res.exec_stage_ms(); res.close(); t2 = time.perf_counter()
    print(f"zkir_exec 2^20 rows: run {1e3*(t1-t0):.2f} ms, free {1e3*(t2-t1):.2f} ms", {k: round(v, 2) for k, v in st.items()})
