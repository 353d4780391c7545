#!/usr/bin/env python3
"""End-to-end throughput of INDEPENDENT runs on one GPU, everything included (zkir_amd/service.py): host interpretation, H2D
upload of the delta log, K1 trace fill and the full proof, with producer threads overlapping the GPU.

usage: pipeline_throughput.py [log2_rows=20] [runs=48] [producers=3]
(verification of the pipelined proofs against the oracle lives in tests/test_gpu_service.py)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkir_amd import runtime as rt, service, spec, stark

args = [a for a in sys.argv[1:] if not a.startswith("--")]
k = int(args[0]) if len(args) > 0 else 20
runs = int(args[1]) if len(args) > 1 else 48
n_prod = int(args[2]) if len(args) > 2 else 3
job = (spec.fib_endless_program().to_bytes(), [], rt.VMConfig(max_cycles=1 << k, enable_execution_trace=True))
ctx = stark.StarkContext(k)
service.prove_many([job] * 2, k, producers=1, ctx=ctx, keep_proofs=False)          # warm-up: workspace, block pool, clocks
rep = service.prove_many([job] * runs, k, producers=n_prod, ctx=ctx, keep_proofs=False)
print(f"{runs} independent 2^{k}-row runs, {n_prod} producer threads: {rep.wall_s * 1e3:.1f} ms wall = {rep.ms_per_run:.2f} ms per proven run "
      f"= {rep.rows_per_s / 1e6:.1f} M rows/s proven end to end (host + PCIe + GPU); per run: interpret {rep.interpret_s / runs * 1e3:.1f} ms, "
      f"upload {rep.upload_s / runs * 1e3:.1f} ms (overlapped across threads)")
