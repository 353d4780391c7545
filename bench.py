#!/usr/bin/env python3
"""bench.py — trace rows/sec of the MI355X execution-trace path on the BASELINE.json workload.

A "step" = one pass of the device hot path over one resident delta log: K1 `trace_fill`
(zkir_amd/csrc/trace_fill.hip) expanding the 2^k-cycle Fibonacci run into the 372 B/row SoA trace.
The delta log (events, tile index, pc/instruction columns) is resident in HBM before the timed region;
the host interpreter that produced it is timed separately and reported as `host_interpret_rows_per_s`.

N ranks = N row shards of one (N * 2^k)-cycle run (weak scaling, no data-path collective: each rank
fills its own row range from its own register snapshot).  Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2-rows", type=int, default=20, help="rows per GPU = 2^k (BASELINE configs[1] = 20)")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from zkir_amd import pipeline as pl, runtime as rt, spec

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    rows_per_gpu = 1 << args.log2_rows
    total_rows = rows_per_gpu * world
    blob = spec.fib_endless_program().to_bytes()

    # ---- host stage (untimed for `value`; reported separately) ------------------------------------
    t0 = time.perf_counter()
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=total_rows, enable_execution_trace=True), tile_rows=args.tile_rows)
    host_s = time.perf_counter() - t0
    assert log.n_rows == total_rows and log.halt_reason == rt.HaltReason.CycleLimit()
    shard = log.shard(rank * rows_per_gpu, (rank + 1) * rows_per_gpu) if world > 1 else log
    t0 = time.perf_counter()
    ddl = pl.upload(shard)
    torch.cuda.synchronize()
    h2d_s = time.perf_counter() - t0
    trace = pl.DeviceTrace(ddl)
    fill_args = pl.trace_fill_args(ddl, trace)
    step_bytes = pl.trace_fill_bytes(ddl)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()                       # untimed pre-warm: let the GPU clocks settle (DVFS) before the W warmup steps
    while time.perf_counter() - t_pre < 0.3:
        for _ in range(20):
            pl.trace_fill(fill_args)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        pl.trace_fill(fill_args)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        pl.trace_fill(fill_args)          # launched on torch's current stream, the one the events are recorded on
        b.record()
    barrier()
    wall = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # ---- parity spot check outside the timed region (full parity lives in tests/ -m gpu) ----------
    n_chk = min(4096, rows_per_gpu)
    got = trace.registers[:, :n_chk].cpu().numpy().view(np.uint64)
    ev_np = shard.reg_events
    for r in (1, 2, 3, 4):
        e = ev_np[ev_np["reg"] == r]
        pos = np.searchsorted(e["vis"], np.arange(n_chk), side="right") - 1
        assert np.array_equal(got[r], e["value"][pos]), "bench parity spot-check failed"

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        value = total_rows * args.steps / wall
        achieved = step_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None                                 # HBM bytes/launch from the committed rocprofv3 PMC passes of this exact workload
        pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_k{args.log2_rows}_t{ddl.tile_rows}.json")
        if os.path.exists(pmc):
            d = json.load(open(pmc))
            traffic = next(iter(d.values())).get("hbm_bytes_per_launch")
        out = {
            "metric": "trace rows/sec (2^20-cycle fib, device-resident delta log -> 372 B/row SoA execution trace)",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"fib_endless 2^{args.log2_rows} cycles per GPU (v3.4 fib loop of tests/cross_module.rs:145-164, max_cycles halt), "
                                   "VMConfig{enable_execution_trace}, stage = K1 trace_fill",
                       "rows_per_gpu": rows_per_gpu, "tile_rows": ddl.tile_rows, "reg_events_per_gpu": ddl.n_events,
                       "parallelism": f"row-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "trace_fill_kernel", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": step_bytes},
            "host_interpret_rows_per_s": total_rows / host_s,
            "h2d_upload_s": h2d_s,
            "end_to_end_rows_per_s_incl_host_and_pcie": rows_per_gpu / (host_s / world + h2d_s + kernel_ms * 1e-3),
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import api as oracle
            n_cpu = min(total_rows, 1 << 20)
            oracle.time_run(blob, 1 << 12)                      # warm the allocator / page cache
            dt, n = oracle.time_run(blob, n_cpu)
            dtf, nf = oracle.time_run(blob, 1 << 14, faithful=True)
            out["cpu_baseline"] = {"value": n / dt, "unit": "rows/s", "cores": 1, "kind": "port",
                                   "sample": f"oracle (C++ restatement of VM::run, linear mode) on the same fib program, {n} rows in {dt:.2f} s; "
                                             f"faithful O(N^2) mode: {nf} rows in {dtf:.2f} s = {nf / dtf:.0f} rows/s"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
