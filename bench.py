#!/usr/bin/env python3
"""bench.py — trace rows/sec of the MI355X prover hot path on BASELINE.json configs[1]:
"2^20-cycle fib trace: BabyBear NTT/LDE + Poseidon2 Merkle on 1 MI355X".

A "step" = one pass of the device hot path over one resident delta log of a 2^k-cycle Fibonacci run:
    K1 trace_fill (372 B/row SoA trace)  ->  main_trace (Baby Bear columns)  ->  LDE (blow-up 2, coset NTT)
    ->  Poseidon2-12 Merkle commitment of the 2^(k+1) LDE rows.
The delta log (register events, tile index, pc/instruction columns) is resident in HBM before the timed region; the
sequential host interpreter that produced it is timed separately (`host_interpret_rows_per_s`, `zkir_exec_ms`).
`--stage trace_fill` times K1 alone.  The trace / LDE / Merkle stages after K1 have no counterpart in the reference (no prover there).

The JSON line also carries `by_config`: the other single-GPU BASELINE configs measured in the same run, outside the timed region —
configs[2] (2^24-cycle fib: commit step + full proof, per-stage rooflines) and configs[4] (2^22-cycle SHA-256 chain: the five
witness kernels with algorithmic bytes, ms and fraction of the HBM peak).

N ranks = N row shards of one (N * 2^k)-cycle run (weak scaling): each rank fills and commits its own row range from its own
register snapshot with no data-path collective; the per-rank Merkle roots are all-gathered over RCCL (16 B per rank) and every
rank hashes the top levels.  N > 1 defaults to BASELINE configs[3]'s per-GPU share, 2^23 rows per GPU (2^26 rows on 8 GPUs).
Host side at N > 1: rank g executes rows [0, g n) of the run untraced on its own core and traces its own n rows
(zkir_interpret_window) — no shard files, no transport; `multi_gpu_end_to_end` reports the time to root of the sharded run with
the host in the loop and the aggregate of N independent runs (one interpreter each), next to the resident-input `value`.
Launch: `python bench.py` (N=1) or torch.distributed.run with --gpus N.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# Montgomery multiplications in one Poseidon2-12 permutation as the hash kernels run it (poseidon2.h, permute_scaled): 4 per S-box x
# (8 x 12 + 22) S-boxes + 1 per partial round (word 0 on its way to the next S-box; round 4: the eleven passive words are updated by a shift and
# an add, no product — round 3 spent 13 products per partial round) x 22 + the 12 products that bring 8 absorbed values and the 4 carried capacity
# words to the input factor
MONT_MUL_PER_PERM = 4 * (8 * 12 + 22) + 22 + 12
PROVE_STAGES = ["main_trace", "lde", "trace_merkle", "lookup_aux", "quotient_and_merkle", "openings", "deep", "fri", "queries"]
# The integer-ALU bound of the Poseidon2 kernels: sum over instruction classes n_i c_i.  A SIMD has 16 lanes, so a wave64 vector
# instruction occupies it for c = 4 cycles; the full-rate 32-bit set (v_mov / v_add_u32 / v_and ...: scripts/isa_hist.py FULL_RATE) for c = 2 — the
# ratios profiles/r02_ubench_alu.txt measured (4.1-4.5 against 2.2-2.4 at a nominal 2.4 GHz).  n_i = the kernel's instruction mix from the committed ISA
# histogram of the shipped code object (profiles/r*_isa_hist.json, scripts/isa_hist.py) scaled to the DYNAMIC count SQ_INSTS_VALU of the committed
# counter pass (profiles/r*_bench_commit_valu_busy.txt).  The c_i are lower bounds of every measured cost, so achieved / peak <= 1 by construction.
SIMDS, PEAK_CLOCK_HZ = 256 * 4, 2.4e9
MAX_LINE_BYTES = 12_000       # the driver parses ONE line; round 4's grew to 26.8 KB and was not parsed: tests/test_bench_line.py holds this


def _fri_schedule(k):
    """stark_prove.inl fri_schedule: folds per committed layer (1, then 3 at a time down to 2^3 values)."""
    ks, log_m = [], k + 1
    while log_m > 3:
        f = 1 if not ks else min(3, log_m - 3)
        ks.append(f)
        log_m -= f
    return ks


def _prove_stage_table(k, W, pms):
    """Algorithmic HBM bytes (the minimum: every operand read once, every result written once), ms and fraction of the HBM peak of
    the prover stages that follow the commitment.  pms = the 9 stage times of zkir_prove (HIP events)."""
    n2 = 2 << k
    from zkir_amd import stark as _stark
    W = W + _stark.W_AUX                                      # the quotient, the openings and the DEEP combination read the main matrix AND the aux matrix of the lookup argument (air.h W_AUX)
    tree = 16 * (2 * n2 - 1)
    fri = 0
    m = n2
    for f in _fri_schedule(k):
        g = m >> f
        fri += 16 * m + 16 * (2 * g - 1)                      # leaf hash reads the layer, the tree is written
        for _ in range(f):
            fri += 16 * m + 8 * m                             # a binary fold reads m extension values, writes m / 2
            m >>= 1
    rows = {
        "quotient_and_merkle": (4 * W * n2 + 32 * n2) + (32 * n2 + tree),   # constraint evaluation reads the LDE once (row j and row j + 2 are the same bytes), writes the Q block; its tree
        "openings": 4 * W * (n2 // 2) + 32 * n2 + 2 * 16 * n2 + 2 * 16 * n2,  # weights written (2 x 16 B); the trace matrices read on the EVEN half of the coset (round 4: their degree is < N), Q on all of it; weights read once
                                                                             # (the rows skipped share their 128-byte lines with the rows read: the HBM traffic is that of the whole matrix)
        "deep": 4 * W * n2 + 16 * n2 + 2 * 16 * n2 + 16 * n2,               # matrix, Q, two inverse columns read; the codeword written
        "fri": fri,
    }
    out = {}
    for name, nbytes in rows.items():
        ms = float(pms[PROVE_STAGES.index(name)])
        out[name] = {"bytes": int(nbytes), "ms": ms, "achieved_GBs": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None,
                     "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None}
    out["openings"]["note"] = "the 16-byte weights are re-read once per group of four columns (38 groups); L2 / Infinity Cache absorb most of it"
    out["fri"]["note"] = "latency-bound: ~80 sequential Poseidon2 tree levels and seven host round trips, not bytes"
    return out


def _strided_passes(stages: int) -> int:
    """Number of strided NTT launches zkir::lde_run (ntt.hip, run_strided_stages) makes for `stages` radix-2 stages: LDS radix-4 passes of
    10 / 8 / 6 / 4 stages, register passes of 3 / 2, a lone stage only for 1."""
    cnt = 0
    while stages > 0:
        if stages >= 10 and stages != 11:
            take = 10
        elif stages == 11:
            take = 8
        elif stages == 9:
            take = 6
        elif stages == 7:
            take = 4
        elif stages == 5:
            take = 3
        else:
            take = stages
        stages -= take
        cnt += 1
    return cnt


_STAGE_KERNELS = {"trace_fill": ["trace_fill_kernel"], "main_trace": ["main_trace_kernel"], "lde": ["ntt_strided_r4_kernel<false", "lde_middle", "ntt_strided_r4_kernel<true"],
                  "merkle": ["leaf_hash_kernel", "compress_kernel", "subtree_kernel"], "merkle_leaves": ["leaf_hash_kernel"],
                  "merkle_levels": ["compress_kernel", "subtree_kernel"]}


def _profiled_traffic(stage: str):
    """(bytes, file): HBM bytes one step's `stage` moves, from the newest committed rocprofv3 counter passes of this exact command (profiles/
    r*_bench_commit_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE per kernel, corrected as MI355X_MICROARCH.md prescribes, averaged per
    launch).  A stage is several kernels: each kernel's per-launch bytes times its launches per step (its launch count relative to the
    stage's first kernel, which runs once per step)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_commit_pmc_traffic.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    names = _STAGE_KERNELS.get(stage, [])
    ref = next((v for kname, v in d.items() if names and kname.startswith(names[0])), None)
    if not ref or not ref.get("WRITE_SIZE_launches"):
        return None, None
    total = 0.0
    for kname, v in d.items():
        if any(kname.startswith(nm) for nm in names):
            total += v["hbm_bytes_per_launch"] * round(v["WRITE_SIZE_launches"] / ref["WRITE_SIZE_launches"])
    return total, os.path.relpath(files[-1], ROOT)


def _check_traffic_source_is_fresh(path: str):
    """"fresh" when the committed counter file was measured at THIS build's kernel sources: the extract scripts write zkir_amd.build.sources_sha16() into the
    files they make (no git needed: the GPU box has none)."""
    from zkir_amd.build import sources_sha16
    try:
        rec = json.load(open(os.path.join(ROOT, path))).get("_kernels_sha16")
    except (OSError, ValueError):
        return "unchecked (file unreadable)"
    if rec is None:
        return "unchecked (file carries no kernels_sha16)"
    return "fresh" if rec == sources_sha16() else "STALE (kernel sources changed since the counter pass)"


def _profile_sha(path: str):
    """The kernel-source hash a committed profile file says it was measured at (a '# kernels_sha16:' line, or the '_kernels_sha16' key of a JSON file), or None."""
    try:
        if path.endswith(".json"):
            return json.load(open(path)).get("_kernels_sha16")
        for line in open(path):
            if line.startswith("# kernels_sha16:"):
                return line.split(":", 1)[1].strip()
    except (OSError, ValueError):
        pass
    return None


def _newest_profile(pattern: str):
    """The committed profile file to read: the one measured at THIS build's kernel sources if there is one (mtime means nothing after a checkout), else the
    highest round tag by NAME — whose figures the caller then marks stale."""
    import glob
    from zkir_amd.build import sources_sha16
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=lambda f: os.path.basename(f))
    if not files:
        return None
    cur = sources_sha16()
    fresh = [f for f in files if _profile_sha(f) == cur]
    return (fresh or files)[-1]


def _profiled_counters(kernel: str):
    """{launch_us, insts_valu, gui_active, valu_busy, clock_ghz, fresh, file} of `kernel` from the newest committed counter pass, or None."""
    from zkir_amd.build import sources_sha16
    path = _newest_profile("r*_bench_commit_valu_busy.txt")
    if not path:
        return None
    sha = None
    for line in open(path):
        if line.startswith("# kernels_sha16:"):
            sha = line.split(":", 1)[1].strip()
        if line.startswith(kernel):
            f = line[60:].split()                        # the kernel name may contain blanks: the numeric columns start at column 60
            try:
                launches, avg_us, insts, act, gui, busy, ipc = (float(x) for x in f[:7])
            except ValueError:
                return None
            return {"launch_us": avg_us, "insts_valu": insts, "gui_active": gui, "valu_busy": busy / 100.0, "clock_ghz": gui / 8 / (avg_us * 1e-6) / 1e9,
                    "fresh": None if sha is None else sha == sources_sha16(), "file": os.path.relpath(path, ROOT)}
    return None


def _alu_roofline(kernel: str, kernel_ms: float):
    """Instruction-class-weighted VALU bound of `kernel` (see SIMDS above): achieved = its dynamic wave-instruction rate, peak = SIMDS x 2.4 GHz / (sum n_i c_i /
    sum n_i) with the static mix of the shipped code object.  None when the committed ISA histogram / counter pass are missing."""
    hist_path = _newest_profile("r*_isa_hist.json")
    ctr = _profiled_counters(kernel)
    if not hist_path or not ctr:
        return None
    hist = json.load(open(hist_path))
    h = hist.get(kernel)
    if not h:
        return None
    from zkir_amd.build import sources_sha16
    c_avg = h["avg_arch_cycles_per_valu"]
    achieved = ctr["insts_valu"] / (kernel_ms * 1e-3)
    peak = SIMDS * PEAK_CLOCK_HZ / c_avg
    cls = h["classes"]
    fresh = hist.get("_kernels_sha16") == sources_sha16() and ctr["fresh"] is True
    return {"bound": "valu-issue", "achieved": achieved if fresh else None, "peak": peak, "unit": "wave64 VALU instr/s",
            "frac": achieved / peak if fresh else None,          #  an instruction count measured at OTHER kernel sources over this run's time is no bound: withheld
            "frac_if_counters_were_fresh": achieved / peak,
            "wave_instr_per_launch": ctr["insts_valu"], "avg_issue_cycles": c_avg, "static_valu": h["valu"],
            "static_mix": {q: cls.get(q, 0) for q in ("multiplier", "add64", "copy", "full_rate_32", "other_valu")},
            "valu_busy_profiled": ctr["valu_busy"], "clock_ghz_profiled": ctr["clock_ghz"],
            # the same fraction INSIDE the committed counter pass (its own instruction count, its own busy cycles: one run, one clock): <= 1 by construction as well
            "frac_in_counter_pass": ctr["insts_valu"] * c_avg / (SIMDS * ctr["gui_active"] / 8.0),
            "isa_hist": os.path.relpath(hist_path, ROOT), "isa_hist_fresh": hist.get("_kernels_sha16") == sources_sha16(),
            "counters": ctr["file"], "counters_fresh": ctr["fresh"]}


def _root_vs_golden(k, root):
    """True / False: the commitment root of the 2^k-cycle fib run against tests/golden/config_roots.json (computed by the CPU oracle alone, offline:
    tests/golden/make_config_roots.py) — BASELINE configs[1]'s "bit-exact root vs CPU"; None when the fixture has no entry for this size."""
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config_roots.json")))["roots"].get(str(k))
    except OSError:
        return None
    return None if gold is None else bool(gold["root"] == root)


def _host_cpu():
    model, cores = "unknown", os.cpu_count()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "logical_cores": cores}


def _hip_time(f, reps=5, warm=2):
    """Median HIP-event time (ms) of f() on torch's current stream (the stream every launch below is made on)."""
    import numpy as np
    import torch
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def _commit_kernel_table(k, W, fill_bytes, stage_ms, lib, sp):
    """Algorithmic bytes, ms and rooflines of the commit stages for 2^k rows."""
    n = 1 << k
    kernels = {"trace_fill": {"bound": "hbm", "bytes": fill_bytes, "ms": stage_ms["trace_fill"]}}
    if "merkle" in stage_ms or "merkle_leaves" in stage_ms:
        #   main_trace: reads the value / state / cycle / pc / instruction columns (164 B/row; the next-row re-read is served by L2), writes W u32 columns
        #   lde: per column `n_inv` strided inverse passes (8 B/elem over N; one radix-4 pass covers up to ten
        #        stages) + fused middle (4N read + 8N written) + as many strided forward passes (8 B/elem over 2N)
        #   merkle: reads the LDE matrix once, writes 16 B per node; ALU-bound (Poseidon2), bytes given for completeness
        n_inv = _strided_passes(max(k - 10, 0))
        kernels["main_trace"] = {"bound": "hbm", "bytes": (164 + 4 * W) * n, "ms": stage_ms["main_trace"]}
        kernels["lde"] = {"bound": "hbm", "bytes": W * n * (8 * n_inv + 12 + 16 * n_inv), "ms": stage_ms["lde"],
                          # the other floor under this stage: a radix-2 butterfly of 31-bit Montgomery arithmetic costs ~33 SIMD-cycles
                          # per wave64 on gfx950 (profiles/r02_ubench_alu.txt); a column of n rows has 30 n butterflies (inverse over n, forward over 2n)
                          "valu_floor_ms": W * n * 30 * 33.0 / 64 / (1024 * 2.4e9) * 1e3,
                          "note": "HBM traffic equals the algorithmic bytes (PMC); the stage sits on a VALU-issue floor of 31-bit modular butterflies about as high as its "
                                  "HBM floor at the 5 TB/s a copy reaches: see valu_floor_ms"}
        # Merkle: the leaf layer (leaf_hash_kernel: one rate-8 sponge per LDE row, ceil(W/8) permutations; reads the matrix once, writes 16 B per
        # leaf) and the 2N - 1 compressions above it (compress_kernel / subtree_kernel), timed separately when the caller split the stage
        perms_leaf, perms_lvl = 2 * n * (-(-W // 8)), 2 * n - 1
        bytes_leaf, bytes_lvl = (4 * W + 16) * 2 * n, (32 + 16) * (2 * n - 1)
        if "merkle_leaves" in stage_ms:
            parts = [("merkle_leaves", perms_leaf, bytes_leaf, "leaf_hash_kernel"), ("merkle_levels", perms_lvl, bytes_lvl, "compress_kernel + subtree_kernel")]
        else:
            parts = [("merkle", perms_leaf + perms_lvl, bytes_leaf + bytes_lvl, "leaf_hash_kernel + compress_kernel + subtree_kernel")]
        for name, perms, nbytes, kern in parts:
            ms = stage_ms[name]
            kernels[name] = {"bound": "valu-issue", "kernels": kern, "bytes": nbytes, "ms": ms, "poseidon2_perms": perms, "poseidon2_perms_per_s": perms / (ms * 1e-3),
                             "mont_mul_per_s": perms * MONT_MUL_PER_PERM / (ms * 1e-3)}
    for v in kernels.values():
        v["achieved_GBs"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        v["frac_of_hbm_peak"] = v["achieved_GBs"] / HBM_PEAK_GBS
    return kernels


def _config2_fib_2p24(lib, sp, k=24):
    """BASELINE configs[2]: 2^24-cycle fib on one GPU — commit step stages + full proof, measured with HIP events.  k = 26: the 2^26
    rows of configs[3] (there row-sharded over 8 GPUs) on ONE device — 25 GB of trace, 122 GB of field matrices, all resident in HBM."""
    import torch
    from zkir_amd import pipeline as pl, runtime as rt, spec, stark
    n = 1 << k
    W = stark.W_MAIN
    blob = spec.fib_endless_program().to_bytes()
    t0 = time.perf_counter()
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
    host_s = time.perf_counter() - t0
    ddl = pl.upload(log); trace = pl.DeviceTrace(ddl); fa = pl.trace_fill_args(ddl, trace)
    ctx = stark.StarkContext(k)
    m = torch.empty((W // 8, n, 8), dtype=torch.int32, device="cuda")        # B8 layout (include/zkir_amd.h)
    L = torch.empty((W // 8, 2 * n, 8), dtype=torch.int32, device="cuda")
    tree = torch.empty(4 * (4 * n - 1), dtype=torch.int32, device="cuda")
    stages = [("trace_fill", lambda: pl.trace_fill(fa)),
              ("main_trace", lambda: pl._check(lib.zkir_main_trace_launch(C.byref(trace.c), n, 0, m.data_ptr(), sp()))),
              ("lde", lambda: pl._check(lib.zkir_lde_launch(ctx.handle, m.data_ptr(), W, L.data_ptr(), sp()))),
              ("merkle_leaves", lambda: pl._check(lib.zkir_merkle_leaves_launch(ctx.handle, L.data_ptr(), W, 2 * n, tree.data_ptr(), sp()))),   # leaf_hash_kernel alone
              ("merkle_levels", lambda: pl._check(lib.zkir_merkle_cap_launch(ctx.handle, tree.data_ptr(), 2 * n, sp())))]                           # together = zkir_merkle_commit_launch
    for _, f in stages:
        f()
    stage_ms = {name: _hip_time(f, reps=3 if k <= 24 else 2, warm=0) for name, f in stages}
    for _, f in stages:                               # the LDE uses its input as scratch: one more pass in order, so the root is that of the trace
        f()
    root = tree[-4:].cpu().numpy().view("uint32").tolist()
    kernels = _commit_kernel_table(k, W, pl.trace_fill_bytes(ddl), stage_ms, lib, sp)
    del m, L, tree
    torch.cuda.empty_cache()
    prove_ms, pms, proof = None, None, None
    pub = rt.public_inputs(log, blob)
    for _ in range(2):                                # first call allocates the context's workspace
        t0 = time.perf_counter()
        proof, pms = stark.prove(ctx, trace, pub, want_stage_ms=True)
        prove_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    verdict = rt.verify_io(proof, pub, [], log.outputs, log.halt_reason)
    verify_ms = (time.perf_counter() - t0) * 1e3
    assert verdict == 0, f"bench: the 2^{k}-row proof was rejected (check {verdict})"
    step_ms = sum(stage_ms.values())
    # the drop-in call at this size: zkir_exec = host interpretation with the log upload and K1 streamed underneath (PCIe inclusive)
    exec_ms = None
    if k <= 24:
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            r_ = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()
            ts.append((time.perf_counter() - t0) * 1e3)
            r_.close()
        exec_ms = min(ts[1:])
    out = {"workload": f"fib_endless 2^{k} cycles, 1 GPU: trace fill + main trace + LDE + Poseidon2 Merkle (commit step), then the full proof "
                       "(AIR quotient, openings, DEEP, FRI, queries)" + ("; the row count of configs[3] (2^26, there over 8 GPUs) on ONE device" if k == 26 else ""),
           "rows": n, "commit_step_ms": step_ms, "commit_rows_per_s": n / (step_ms * 1e-3), "stage_ms": stage_ms, "roofline_by_stage": kernels,
           "prove_ms": prove_ms, "prove_stage_ms": dict(zip(PROVE_STAGES, pms)), "prove_rows_per_s": n / (prove_ms * 1e-3),
           "prove_stage_roofline": _prove_stage_table(k, W, list(pms)),
           "proof_bytes": int(len(proof) * 4), "verify_ms_host": verify_ms, "merkle_root": root, "proof_trace_root_matches_commit": stark.trace_root(proof) == root,
           "host_interpret_s": host_s, "hbm_resident_GB": (372 * n + (12 * W + 440) * n) / 1e9,
           "zkir_exec_ms": exec_ms, "zkir_exec_rows_per_s": n / (exec_ms * 1e-3) if exec_ms else None}
    ctx.close(); log.close()
    del trace, ddl
    torch.cuda.empty_cache()
    return out


def _config4_sha_2p22(lib, sp):
    """BASELINE configs[4]: SHA-256 hash-chain program for 2^22 cycles — the syscall-chip trace columns (witness.hip kernels)."""
    import hashlib
    import numpy as np
    import torch
    from zkir_amd import pipeline as pl, runtime as rt, spec
    k = 22
    n = 1 << k
    dev = torch.device("cuda")
    t0 = time.perf_counter()
    log = rt.interpret(spec.sha256_chain_program().to_bytes(), [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
    host_s = time.perf_counter() - t0
    n_ops, n_blk = len(log.mem_events), len(log.sha_blocks)
    ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); fa = pl.trace_fill_args(ddl, tr)
    ev = pl._to_dev(log.mem_events, dev)
    rows_c, sort_c = pl.MemopColumns(n_ops, dev), pl.MemopColumns(n_ops, dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev); scratch = torch.empty(n, dtype=torch.uint8, device=dev)
    blk = pl._to_dev(log.sha_blocks, dev)
    sha_stride = (n_blk + 63) // 64 * 64                      # a multiple of 4: the 16-byte-store chip kernel
    out = torch.empty((608, sha_stride), dtype=torch.int32, device=dev); ts = torch.empty(n_blk, dtype=torch.int64, device=dev)
    kern = {}

    def rec(name, f, nbytes, note=None):
        ms = _hip_time(f)
        kern[name] = {"bound": "hbm", "bytes": int(nbytes), "ms": ms, "achieved_GBs": nbytes / (ms * 1e-3) / 1e9,
                      "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if note:
            kern[name]["note"] = note
    rec("trace_fill", lambda: pl.trace_fill(fa), pl.trace_fill_bytes(ddl))
    rec("memops_expand_csr", lambda: pl._check(lib.zkir_memops_expand_csr_launch(ev.data_ptr(), n_ops, n, 0, C.byref(rows_c.c), offs.data_ptr(), scratch.data_ptr(), sp())),
        (24 + 39) * n_ops + 8 * (n + 1) + n,
        "TraceRow.memory_ops in row order + the CSR row offsets + the per-row shape flags of the sort, ONE pass over the events (round 2: three passes; "
        "the per-row binary search alone moved 12x its algorithmic bytes)")
    rec("memops_sort", lambda: pl._check(lib.zkir_memops_sort_prepared_launch(ev.data_ptr(), n_ops, 0, offs.data_ptr(), scratch.data_ptr(), C.byref(sort_c.c), sp())),
        (24 + 39) * n_ops + n, "= ExecutionResult::get_memory_trace(): rank in a two-run merge per row (keys of the touched segments staged in LDS)")
    rec("sha256_chip", lambda: pl._check(lib.zkir_sha256_chip_launch(blk.data_ptr(), n_blk, out.data_ptr(), sha_stride, ts.data_ptr(), sp())), (72 + 2432 + 8) * n_blk)
    o = out[:, [0, n_blk // 2, n_blk - 1]].cpu().numpy().view(np.uint32)       # parity spot-check outside the timings: digests vs hashlib
    for col, idx in enumerate([0, n_blk // 2, n_blk - 1]):
        msg = log.sha_blocks[idx]["message_block"].astype(">u4").tobytes()[:32]
        assert o[600:608, col].astype(">u4").tobytes() == hashlib.sha256(msg).digest(), "bench: SHA chip spot-check failed"
    res = {"workload": "sha256_chain_program 2^22 cycles, 1 GPU (pattern of zkir-runtime/tests/crypto_edge_cases.rs:405-427): K1 + memory-op CSR/expand/sort + SHA-256 chip",
           "rows": n, "memory_ops": n_ops, "sha256_blocks": n_blk, "reg_events": ddl.n_events, "kernels": kern,
           "witness_ms_total": sum(v["ms"] for v in kern.values()), "host_interpret_s": host_s}
    log.close()
    return res


def _cpu_baseline(blob, k, commit):
    """The reference path's own quantity — VM execution-trace rows/s — from the CPU oracle (C++ restatement of VM::run; the Rust
    reference cannot be built here), one thread, on this box's host cores.  Bounded to ~10-30 s."""
    from oracle import api as oracle
    oracle.time_run(blob, 1 << 12)                      # warm the allocator / page cache
    n_cpu = 1 << min(k, 22)
    dt, nn = oracle.time_run(blob, n_cpu)
    kf = 17                                             # faithful mode is O(N^2): 2^17 rows take a few seconds, 2^18 four times that
    dtf, nf = oracle.time_run(blob, 1 << kf, faithful=True)
    cpu = _host_cpu()
    sample = (f"oracle/zkir_oracle.cpp (C++ -O3 restatement of VM::run, vm.rs:208-358) on {nn} rows of the same fib program, linear mode "
              f"(the reference's per-cycle O(N) memory-trace filter, vm.rs:287-298, replaced by a cursor): {dt:.3f} s = {nn / dt:.3g} rows/s; "
              f"faithful O(N^2) mode: {nf} rows in {dtf:.2f} s = {nf / dtf:.3g} rows/s (the largest size that finishes in seconds); "
              f"host CPU: {cpu['model']}, {cpu['logical_cores']} logical cores, 1 used")
    out = {"value": nn / dt, "unit": "rows/s", "cores": 1, "kind": "port", "sample": sample,
           "linear_rows_per_s": nn / dt, "linear_rows": nn, "faithful_rows_per_s": nf / dtf, "faithful_rows": nf, "host_cpu": cpu}
    if k == 20:
        # north_star's target is stated at 2^24 cycles: time the linear mode AT that size, ~2-4 s.  The oracle keeps its rows
        # (372 B each: 6.2 GB, ~9 GB at the peak of the vector's growth); a host without that much free memory builds every row without keeping it.
        avail = 0
        try:
            for line in open("/proc/meminfo"):
                if line.startswith("MemAvailable"):
                    avail = int(line.split()[1]) * 1024
        except OSError:
            pass
        keep = avail >= (20 << 30)
        dt24, n24 = oracle.time_run(blob, 1 << 24, build_only=not keep)
        out["linear_2p24"] = {"rows": n24, "seconds": dt24, "rows_per_s": n24 / dt24, "rows_kept": keep,
                              "what": "the same oracle, linear mode, one thread, 2^24 cycles of the same fib program" + ("" if keep else " (rows built but not stored: host memory)")}
    if commit:                                          # self-defined stages: NOT the reference path; a separate, labelled figure
        from oracle import stark_api as so
        threads = max(1, min(cpu["logical_cores"] or 1, 64))
        rows = oracle.run(blob, max_cycles=1 << k, enable_execution_trace=True).rows
        t0 = time.perf_counter()
        m = so.to_committed(so.main_trace(rows, so.public_inputs(len(rows), blob)))      # the columns the commit stage works on (default mode: 152)
        t_main = time.perf_counter() - t0
        del rows
        from bench_cpu import api as cpu_port
        root, t_lde, t_merkle = cpu_port.commit_port(m, threads)
        out["commit_stage_self_defined"] = {
            "rows": 1 << k, "threads": threads, "main_trace_s_1_thread": t_main, "lde_s": t_lde, "merkle_s": t_merkle,
            "rows_per_s": (1 << k) / (t_lde + t_merkle), "merkle_root": [int(x) for x in root],
            "what": "the SAME commit stage at the SAME size on this box's host cores: bench_cpu/cpu_commit_port.cpp (Montgomery arithmetic, radix-2 NTTs with "
                    "twiddle tables, std::thread over columns / leaves); stages absent from the reference (self-defined) — a labelled side figure, "
                    "never part of `value`; its root must equal the GPU's (checked below)"}
    return out


def _pick(d, *keys):
    return {q: d[q] for q in keys if isinstance(d, dict) and q in d}


def headline(full: dict) -> dict:
    """The ONE line the driver parses (VERDICT r4 task 1): the contract's keys + roofline + cpu_baseline + the proof figures, numbers only; everything else
    (by_config, prove_by_mode, roofline_by_stage, pipelined_*, prose) stays in bench_detail.json.  Pure function of the full record: tests/test_bench_line.py."""
    line = _pick(full, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    cfg = full.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:300], **_pick(cfg, "rows_per_gpu", "main_trace_width", "parallelism")}
    rf = dict(full.get("roofline") or {})
    alu = rf.get("alu")
    if isinstance(alu, dict):
        rf["alu"] = _pick(alu, "bound", "achieved", "peak", "unit", "frac", "wave_instr_per_launch", "avg_issue_cycles", "static_mix", "valu_busy_profiled",
                          "clock_ghz_profiled", "frac_in_counter_pass", "isa_hist_fresh", "counters_fresh")
    line["roofline"] = _pick(rf, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source_freshness", "algorithmic_bytes_per_launch", "kernel_ms", "alu")
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, "value", "unit", "cores", "kind", "linear_rows", "linear_rows_per_s", "faithful_rows", "faithful_rows_per_s")
        c["sample"] = str(cb.get("sample", ""))[:200]
        if isinstance(cb.get("linear_2p24"), dict):
            c["linear_2p24"] = _pick(cb["linear_2p24"], "rows", "seconds", "rows_per_s")
        c["host_cpu"] = (cb.get("host_cpu") or {}).get("model")
        line["cpu_baseline"] = c
    line.update(_pick(full, "prove_ms", "prove_ms_mode2", "prove_ms_100bit", "prove_stage_ms", "proof_bytes", "merkle_root", "merkle_root_equals_oracle_golden",
                      "gpu_ms_per_step_hip_events", "hbm_copy_GBs_measured", "stage_ms", "zkir_exec_rows_per_s"))
    t = full.get("target_10x_at_2p24")
    if isinstance(t, dict):
        line["target_10x_at_2p24"] = _pick(t, "ratio", "gpu_path_rows_per_s", "cpu_linear_rows_per_s")
    c2 = (full.get("by_config") or {}).get("configs[2]")
    if isinstance(c2, dict):
        line["config2_2p24"] = _pick(c2, "rows", "commit_rows_per_s", "prove_ms", "proof_bytes")
    if (full.get("n_gpus") or 1) > 1 or full.get("process_group"):
        line.update(_pick(full, "merkle_roots_all_ranks", "allgather_cap_ms", "per_gpu_local_stage_rows_per_s", "efficiency_vs_same_size_single_gpu",
                          "end_to_end_rows_per_s_incl_host"))
        pg = full.get("process_group")
        if isinstance(pg, dict):
            line["process_group"] = _pick(pg, "backend", "world_size")
        sp_ = full.get("segment_prove")
        if isinstance(sp_, dict):
            line["segment_prove"] = _pick(sp_, "segments", "ms_all_segments_in_parallel", "rows_per_s_proven", "verify_chain_code", "error")
    line["detail"] = "bench_detail.json"
    return line


def emit(full: dict):
    """Full record -> bench_detail.json (next to this script; also under gpurun_out/ so it comes back from a GPU box) and stderr; the headline -> ONE stdout line."""
    detail = json.dumps(full)
    for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(detail + "\n")
        except OSError:
            pass
    print("bench_detail: " + detail, file=sys.stderr, flush=True)
    text = json.dumps(headline(full))
    assert len(text) < MAX_LINE_BYTES, f"bench line is {len(text)} bytes (limit {MAX_LINE_BYTES})"
    print(text, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log2-rows", type=int, default=None, help="rows per GPU = 2^k (default: 20 = BASELINE configs[1] at N=1; 23 = configs[3]'s "
                    "per-GPU share, 2^26 rows over 8 GPUs, at N>1)")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--stage", choices=["commit", "trace_fill"], default="commit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prove", action="store_true", help="skip the end-to-end proof / pipelined sections that follow the timed region "
                    "(profiling runs: keeps the rocprofv3 per-kernel averages to the kernels of the timed steps)")
    ap.add_argument("--no-by-config", action="store_true", help="skip the configs[2] / configs[4] sections (profiling runs)")
    ap.add_argument("--only-config", choices=["2", "3", "4"], default=None, help="run only that by_config section and print it (profiling runs; 3 = 2^26 rows on one GPU)")
    args = ap.parse_args()

    # Read by the HSA runtime when it initialises (the first HIP call of the process), so it has to be in the environment BEFORE torch
    # touches the device: the host driver only supports dmabuf IPC, and RCCL's intra-node transport needs it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist

    from zkir_amd import pipeline as pl, runtime as rt, spec, stark

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if world & (world - 1):
        raise SystemExit(f"bench.py: --gpus {world} is not a power of two: the row-sharded commitment is a binary Merkle tree over the "
                         "per-GPU subtree roots (zkir_merkle_cap_launch), so the number of row shards must be 1, 2, 4 or 8")
    # Functional test of the N > 1 path on a 1-GPU box: ZKIR_BENCH_BACKEND=gloo ZKIR_BENCH_SHARE_GPU=1 runs every rank on
    # cuda:0 and stages the 16-byte root exchange through the host.  Production (the driver's launch): nccl = RCCL, one GPU per rank.
    backend = os.environ.get("ZKIR_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("ZKIR_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    # ZKIR_BENCH_FORCE_DIST=1: take the N > 1 code path at ANY world size — at world 1 (torchrun --nproc-per-node 1, a one-GPU box) that is the
    # window interpretation, the device all-gather of the root over RCCL, the cap, the f64 all-reduces, the barriers, the object collectives
    # and the segment proofs + zkir_verify_chain, all on a real `nccl` process group (tests/test_gpu_multirank.py::test_world1_over_rccl)
    dist_mode = world > 1 or os.environ.get("ZKIR_BENCH_FORCE_DIST") == "1"
    if dist_mode:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    lib = rt.lib()
    sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    if args.only_config:
        torch.zeros(1 << 20, device="cuda").sum().item()
        fn = {"2": lambda: _config2_fib_2p24(lib, sp), "3": lambda: _config2_fib_2p24(lib, sp, 26), "4": lambda: _config4_sha_2p22(lib, sp)}[args.only_config]
        print(json.dumps({"by_config": {("configs[3] on one GPU" if args.only_config == "3" else f"configs[{args.only_config}]"): fn()}}))
        return

    k = args.log2_rows if args.log2_rows is not None else (20 if not dist_mode else 23)
    n = 1 << k
    total_rows = n * world
    W = stark.W_MAIN
    blob = spec.fib_endless_program().to_bytes()

    # ---- host stage (untimed for `value`; reported separately) ------------------------------------
    # Execution is sequential.  N = 1: one interpretation, in-process.  N > 1: EVERY rank executes the rows before its shard untraced and
    # traces its own (zkir_interpret_window, below): no shard files, nothing travels between the processes.
    host_first_s = host_s = None
    per_rank_host_s = None
    log = None
    if not dist_mode:
        t0 = time.perf_counter()
        log = rt.interpret(blob, [], rt.VMConfig(max_cycles=total_rows, enable_execution_trace=True), tile_rows=args.tile_rows)
        host_first_s = host_s = time.perf_counter() - t0
        assert log.n_rows == total_rows and log.halt_reason == rt.HaltReason.CycleLimit()
        # steady state: the log buffers of a finished run are recycled (host.h block pool), so later runs do not page-fault their way
        # through ~50 B/row of fresh memory; time a second run and give its buffers back
        for _ in range(2):
            t0 = time.perf_counter()
            rt.interpret(blob, [], rt.VMConfig(max_cycles=total_rows, enable_execution_trace=True), tile_rows=args.tile_rows).close()
            host_s = min(host_s, time.perf_counter() - t0)
        shard = log
    else:
        # Execution is sequential (vm.rs:208-358): the state at row g*n exists only once rows [0, g*n) have been executed.  Rank g
        # executes them itself, UNTRACED (zkir_interpret_window: no log stores, 1.5-2x the traced rate), then traces its own rows —
        # so GPU g has its shard after g*n*t_untraced + n*t_traced, sooner than a single traced interpreter on rank 0 would reach
        # those rows, and nothing travels between the processes (round 2 wrote every shard to /dev/shm as .npz and read it back).
        # The window also covers the segment-proof shard of the rank ([g(n-1), g(n-1)+n): overlaps the previous rank's by one row).
        want_segments = args.stage == "commit" and not args.no_prove
        vm_cfg = rt.VMConfig(max_cycles=total_rows, enable_execution_trace=True)
        lo = rank * (n - 1) if want_segments else rank * n
        dist.barrier()
        t0 = time.perf_counter()
        win = rt.interpret(blob, [], vm_cfg, tile_rows=args.tile_rows, window=(lo, (rank + 1) * n))
        host_first_s = host_s = time.perf_counter() - t0                  # this rank: fast-forward + its traced window
        assert win.n_rows == (rank + 1) * n - lo and win.cycle_base == lo
        shard = win.shard(rank * n, (rank + 1) * n)
        seg_shards = []
        run_pub_bytes = [None]
        if want_segments:
            seg_shards.append(win.shard(rank * (n - 1), rank * (n - 1) + n))
            if rank == world - 1:                                         # the G rows the G overlapping segments leave at the end of the run
                seg_shards.append(win.shard(world * (n - 1), total_rows))
                assert not win.window_open and win.cycles == total_rows and win.halt_reason == rt.HaltReason.CycleLimit()
                run_pub_bytes[0] = bytes(rt.public_inputs(win, blob))      # the last rank executed the run to its halt: outputs / halt reason / cycles are the run's
            dist.broadcast_object_list(run_pub_bytes, src=world - 1)
        per_rank_host_s = [None] * world
        dist.all_gather_object(per_rank_host_s, host_s)
        host_first_s = host_s = max(per_rank_host_s)                      # the last rank's: it executes the whole run
        win.close()
    torch.zeros(1 << 20, device="cuda").sum().item()   # HIP context / allocator warm-up is not part of the H2D figure
    # The delta log sits in pinned pool blocks (host.h: Buf / block_acquire), so the copies are plain DMA.  The FIRST multi-megabyte host-to-device copy of a process costs
    # ~7 ms whatever the memory is (scripts/time_upload.py, profiles/r06_pinned_delta_log.txt: the runtime sets its copy path up), so the upload is timed twice and the
    # second one is the figure; the first is reported beside it.
    t0 = time.perf_counter()
    ddl = pl.upload(shard)
    torch.cuda.synchronize()
    h2d_first_s = time.perf_counter() - t0
    del ddl
    t0 = time.perf_counter()
    ddl = pl.upload(shard)
    torch.cuda.synchronize()
    h2d_s = time.perf_counter() - t0
    trace = pl.DeviceTrace(ddl)
    fill_args = pl.trace_fill_args(ddl, trace)
    commit = args.stage == "commit"
    if commit:
        ctx = stark.StarkContext(k)
        m = torch.empty((W // 8, n, 8), dtype=torch.int32, device="cuda")    # B8 layout (include/zkir_amd.h)
        L = torch.empty((W // 8, 2 * n, 8), dtype=torch.int32, device="cuda")
        tree = torch.empty(4 * (4 * n - 1), dtype=torch.int32, device="cuda")

    stages = [("trace_fill", lambda: pl.trace_fill(fill_args))]
    if commit:
        stages += [("main_trace", lambda: pl._check(lib.zkir_main_trace_launch(C.byref(trace.c), n, 0, m.data_ptr(), sp()))),
                   ("lde", lambda: pl._check(lib.zkir_lde_launch(ctx.handle, m.data_ptr(), W, L.data_ptr(), sp()))),
                   ("merkle_leaves", lambda: pl._check(lib.zkir_merkle_leaves_launch(ctx.handle, L.data_ptr(), W, 2 * n, tree.data_ptr(), sp()))),   # leaf_hash_kernel alone
                   ("merkle_levels", lambda: pl._check(lib.zkir_merkle_cap_launch(ctx.handle, tree.data_ptr(), 2 * n, sp())))]                           # together = zkir_merkle_commit_launch

    # N > 1: the step ends with the only collective of the path — an all-gather of the per-GPU subtree roots (16 B per rank over
    # RCCL/xGMI) — and every rank hashes the top log2(G) levels over them (zkir_merkle_cap_launch), so `value` is the rate of the
    # complete G-GPU commitment, not of G unrelated ones.
    gathered, cap_root = None, [None]
    if commit and dist_mode:
        gathered = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(world)]

        def exchange():
            if backend == "nccl":
                dist.all_gather(gathered, tree[-4:])
            else:
                host = [torch.zeros(4, dtype=torch.int32) for _ in range(world)]
                dist.all_gather(host, tree[-4:].cpu())
                for dst, src in zip(gathered, host):
                    dst.copy_(src)
            cap_root[0] = stark.merkle_cap(ctx, torch.stack(gathered))
        stages.append(("allgather_cap", exchange))

    def step():
        for _, f in stages:
            f()

    def barrier():
        if dist_mode:
            dist.barrier()
        torch.cuda.synchronize()

    t_pre = time.perf_counter()                       # untimed pre-warm: let the GPU clocks settle (DVFS) before the W warmup steps
    if not dist_mode:
        while time.perf_counter() - t_pre < 0.3:
            step()
            torch.cuda.synchronize()
    else:                                             # the step contains a collective: every rank must run the same number of them
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events between the stages INSIDE the timed region, on the stream the kernels are launched on (torch's current stream):
    # the per-stage durations the rooflines are computed from are those of the timed steps themselves.
    marks = [[torch.cuda.Event(enable_timing=True) for _ in range(len(stages) + 1)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        for j, (_, f) in enumerate(stages):
            marks[i][j].record()
            f()
        marks[i][-1].record()
    barrier()
    wall = time.perf_counter() - t0
    gpu_ms_per_step = marks[0][0].elapsed_time(marks[-1][-1]) / args.steps
    if dist_mode:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    stage_ms = {name: float(np.mean([marks[i][j].elapsed_time(marks[i][j + 1]) for i in range(args.steps)])) for j, (name, _) in enumerate(stages)}

    # ---- measured HBM copy bandwidth of this device (report the measured peak next to the nominal 8 TB/s): the library's 16-byte-per-lane grid-stride copy
    #      (zkir_amd_experimental.h; MI355X_MICROARCH.md quotes ~6.3 TB/s for it — torch's copy_ reached 4.76: VERDICT r4 weak #6)
    lib.zkir_hbm_copy_peak_gbs.restype = C.c_double
    lib.zkir_hbm_copy_peak_gbs.argtypes = [C.c_uint64, C.c_void_p]
    hbm_copy_gbs = max(float(lib.zkir_hbm_copy_peak_gbs(1 << 30, sp())) for _ in range(3))

    # ---- end-to-end prove (BASELINE metric's "end-to-end prove ms"): AIR quotient + openings + DEEP + FRI on top of the commit ----
    prove_ms, prove_stage_ms, proof_bytes, verify_ms, prove_ms_100bit = None, None, None, None, None
    if commit and not dist_mode and not args.no_prove:   # a proof is for the whole run (its AIR pins cycle[0] = 0): single-GPU only
        pub = rt.public_inputs(log, blob)
        for _ in range(3):                            # first call allocates the context's workspace
            t0 = time.perf_counter()
            proof, pms = stark.prove(ctx, trace, pub, want_stage_ms=True)
            prove_ms = (time.perf_counter() - t0) * 1e3
        prove_stage_ms = dict(zip(PROVE_STAGES, pms))
        proof_bytes = int(len(proof) * 4)
        t0 = time.perf_counter()
        assert rt.verify_io(proof, pub, [], log.outputs, log.halt_reason) == 0, "bench: proof (or its I/O / halt claim) rejected by zkir_verify_io"
        verify_ms = (time.perf_counter() - t0) * 1e3
        # the same proof at the second parameter set (zkir_prover_params: 84 FRI queries + 16 grinding bits = 100 conjectured bits of FRI soundness at blow-up 2;
        # the capacity-4 Poseidon2 sponge still caps collision resistance at ~62 bits: README)
        pub100 = rt.public_inputs(log, blob, num_queries=84, pow_bits=16)
        for _ in range(3):
            t0 = time.perf_counter()
            proof100 = stark.prove(ctx, trace, pub100)
            prove_ms_100bit = (time.perf_counter() - t0) * 1e3
        assert rt.verify(proof100, pub100) == 0 and int(proof100[4]) == 84 and int(proof100[6]) == 16

    # ---- the opt-in proof MODES of round 4 at the same size: the fib run in mode 2 (+ the I/O argument), and the array loop of spec.memory_loop_program
    #      (4 of 13 rows are loads / stores) in modes 0, 2 and 3 (+ the memory argument; the memory witness made on the device) — what saying more costs
    prove_by_mode = prove_ms_mode2 = None
    if commit and not dist_mode and not args.no_prove and k <= 22:
        try:
            def timed_prove(tr_, pub_, reps=5):
                best, pr_, st_ = None, None, None
                stark.prove(ctx, tr_, pub_)                    # (a wider mode grows the context's workspace on its first call)
                all_ms.clear()
                for _ in range(reps):
                    t0 = time.perf_counter()
                    pr, st = stark.prove(ctx, tr_, pub_, want_stage_ms=True)
                    dt = (time.perf_counter() - t0) * 1e3
                    all_ms.append([round(dt, 2), round(float(sum(st)), 2)])      # (wall, sum of the stage times: the difference is host time outside the stages)
                    if best is None or dt < best:
                        best, pr_, st_ = dt, pr, st            # the stage times of the run that is reported
                return best, pr_, st_
            prove_by_mode = {}
            all_ms = []
            ms2, pr2, _ = timed_prove(trace, rt.public_inputs(log, blob, [], io_mode=True))
            assert rt.verify(pr2) == 0
            prove_ms_mode2 = ms2
            prove_by_mode["fib, mode 2 (+ I/O argument, 160 + 48 columns)"] = {"prove_ms": ms2, "proof_bytes": int(len(pr2) * 4), "vs_mode_0": ms2 / prove_ms}
            mblob = spec.memory_loop_program(min(65535, n // 13)).to_bytes()
            mlog = rt.interpret(mblob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
            mddl = pl.upload(mlog); mtr = pl.DeviceTrace(mddl); pl.trace_fill(pl.trace_fill_args(mddl, mtr)); torch.cuda.synchronize()
            base = None
            for mode, label in ((0, "mode 0 (152 + 40 columns)"), (2, "mode 2 (160 + 48)"), (3, "mode 3 (+ memory argument, bitwise opcodes, shifts, MUL: 264 + 96; witness on the device)")):
                mpub = rt.public_inputs(mlog, mblob, [], io_mode=mode == 2, mem_mode=mode == 3)
                ms, pr, st = timed_prove(mtr, mpub)
                assert rt.verify(pr, mpub) == 0, f"bench: mode-{mode} proof of the array loop rejected"
                base = ms if base is None else base
                prove_by_mode[f"array loop ({mlog.n_rows} rows), {label}"] = {"prove_ms": ms, "stage_ms": dict(zip(PROVE_STAGES, st)), "proof_bytes": int(len(pr) * 4), "vs_mode_0": ms / base, "all_ms": list(all_ms)}
            # (round 6) mode 4 = mode 3 + MULH / DIVU / REMU / DIV / REM + hash syscalls as a tape + the boundary cell: the same array loop (what the wider rows cost a program
            # that uses none of it), the wide-arithmetic loop, and BASELINE configs[4]'s program — the SHA-256 hash chain — proven with every digest recomputed by the verifier
            mpub4 = rt.public_inputs(mlog, mblob, [], wide_mode=True, mem_witness="device")
            ms, pr, st = timed_prove(mtr, mpub4)
            assert rt.verify(pr, mpub4) == 0
            prove_by_mode[f"array loop ({mlog.n_rows} rows), mode 4 (288 + 128 columns; witness on the device)"] = {"prove_ms": ms, "stage_ms": dict(zip(PROVE_STAGES, st)), "proof_bytes": int(len(pr) * 4), "vs_mode_0": ms / base}
            for label, prog in (("wide-arithmetic loop (6 of 18 rows MULH / DIVU / REMU / DIV / REM)", spec.wide_loop_program()),
                                ("signed-division loop on raw 64-bit registers (6 of 12 rows wide, about half of them through the verifier-recomputed wide tape)", spec.signed_division_loop_program()),
                                ("SHA-256 hash chain = configs[4]'s program (a hash syscall every 6 rows)", spec.sha256_chain_program())):
                wblob = prog.to_bytes()
                wlog = rt.interpret(wblob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
                wddl = pl.upload(wlog); wtr = pl.DeviceTrace(wddl); pl.trace_fill(pl.trace_fill_args(wddl, wtr)); torch.cuda.synchronize()
                t0 = time.perf_counter(); wpub = rt.public_inputs(wlog, wblob, [], wide_mode=True, mem_witness="host"); t_wit = (time.perf_counter() - t0) * 1e3
                ms, pr, st = timed_prove(wtr, wpub)
                t0 = time.perf_counter(); vrc = rt.verify(pr, wpub); t_ver = (time.perf_counter() - t0) * 1e3
                assert vrc == 0, f"bench: mode-4 proof of the {label} rejected ({vrc})"
                prove_by_mode[f"{label}, {wlog.n_rows} rows, mode 4"] = {"prove_ms": ms, "stage_ms": dict(zip(PROVE_STAGES, st)), "proof_bytes": int(len(pr) * 4), "host_witness_ms": t_wit,
                                                                        "hash_calls": int(wpub._mem_ref.n_hash_calls), "touched_cells": int(wpub._mem_ref.n_cells), "verify_ms_host": t_ver}
                wlog.close(); del wddl, wtr
            t0 = time.perf_counter(); hw = rt.MemcheckWitness(mlog, mblob); t_host = (time.perf_counter() - t0) * 1e3
            prove_by_mode["array loop: memory witness by the host's sequential replay instead (zkir_memcheck_witness_of)"] = {"ms": t_host, "accesses": hw.n_accesses, "cells": hw.n_cells}
            mlog.close()
        except Exception as e:                        # a side figure must not take the bench line down
            prove_by_mode = {"error": repr(e)}

    # ---- N > 1: the run PROVEN, one segment per GPU (no data-path collective: a segment needs its own rows only); the proofs are
    #      gathered on rank 0 and verified there as ONE run (zkir_verify_chain: first state initial, states link, same public inputs)
    segment_prove = None
    if commit and dist_mode and not args.no_prove:
        try:
            run_pub = rt.PublicInputsC.from_buffer_copy(run_pub_bytes[0]).with_program(blob)   # the borrowed program pointer of another process means nothing here
            seg_proofs, seg_ms = [], None
            for si, sh in enumerate(seg_shards):
                ddl2 = pl.upload(sh); tr2 = pl.DeviceTrace(ddl2); pl.trace_fill(pl.trace_fill_args(ddl2, tr2))
                pub2 = rt.PublicInputsC.from_buffer_copy(run_pub); pub2.n_real = sh.n_rows
                ctx2 = ctx if stark.padded_log_n(sh.n_rows) == k else stark.StarkContext(stark.padded_log_n(sh.n_rows))
                if si == 0:
                    stark.prove(ctx2, tr2, pub2)                      # first call allocates the context's workspace
                    barrier()
                    t0 = time.perf_counter()
                    pr, pms = stark.prove(ctx2, tr2, pub2, want_stage_ms=True)
                    barrier()
                    seg_ms = (time.perf_counter() - t0) * 1e3
                else:
                    pr = stark.prove(ctx2, tr2, pub2)
                seg_proofs.append(pr)
                if ctx2 is not ctx:
                    ctx2.close()
                del ddl2, tr2
            tmax = torch.tensor([seg_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            gathered_proofs = [None] * world if rank == 0 else None
            dist.gather_object([p.tobytes() for p in seg_proofs], gathered_proofs, dst=0)
            if rank == 0:
                chain = [np.frombuffer(b, dtype=np.uint32) for per_rank in gathered_proofs for b in per_rank]
                t0 = time.perf_counter()
                rc = rt.verify_chain(chain, run_pub)
                segment_prove = {"segments": len(chain), "rows_per_segment": n, "overlap_rows": 1, "ms_all_segments_in_parallel": float(tmax.item()),
                                 "rows_per_s_proven": total_rows / (float(tmax.item()) * 1e-3), "stage_ms_rank0": dict(zip(PROVE_STAGES, pms)),
                                 "proof_bytes_total": int(sum(len(c) for c in chain) * 4), "verify_chain_code": int(rc), "verify_chain_ms_host": (time.perf_counter() - t0) * 1e3,
                                 "note": "one ZKIR-STARK proof per GPU over its row shard plus the first row of the next (boundary states in the proof header, pinned by "
                                         "the AIR); zkir_verify_chain accepts the segments as one run; no collective in the proving path"}
                assert rc == 0, f"bench: segment chain rejected ({rc})"
        except Exception as e:                                    # the bench line must not depend on this leg
            segment_prove = {"error": repr(e)}

    # ---- N > 1 with the HOST IN THE LOOP: (a) time to root of ONE run row-sharded over the G GPUs, every rank starting
    #      from the program blob: zkir_exec_window (rows before the shard executed untraced on the rank's core, its own rows traced
    #      with upload + K1 streamed underneath) -> main trace -> LDE -> Merkle -> all-gather + cap; (b) G INDEPENDENT runs, one per
    #      GPU, each interpreted by its own rank: the mode whose aggregate scales with G, since one run is bounded by one core
    multi_e2e = None
    if commit and dist_mode:
        try:
            def commit_from(cols, n_rows):
                pl._check(lib.zkir_main_trace_launch(C.byref(cols), n_rows, 0, m.data_ptr(), sp()))
                pl._check(lib.zkir_lde_launch(ctx.handle, m.data_ptr(), W, L.data_ptr(), sp()))
                pl._check(lib.zkir_merkle_commit_launch(ctx.handle, L.data_ptr(), W, 2 * n, tree.data_ptr(), sp()))

            def timed(body, reps=3):
                best = None
                for _ in range(reps):                             # first repetition: cold block pool / device allocations
                    barrier()
                    t0 = time.perf_counter()
                    parts = body()
                    barrier()
                    dt = time.perf_counter() - t0
                    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    if best is None or float(t.item()) < best[0]:
                        best = (float(t.item()), parts)
                return best

            def sharded_run():
                t0 = time.perf_counter()
                res = rt.VM(blob, [], rt.VMConfig(max_cycles=total_rows, enable_execution_trace=True)).run_window(rank * n, (rank + 1) * n)
                t_exec = time.perf_counter() - t0
                commit_from(res.execution_trace.columns, n)
                exchange()
                torch.cuda.synchronize()
                res.close()
                return t_exec

            def independent_run():
                t0 = time.perf_counter()
                res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()
                t_exec = time.perf_counter() - t0
                commit_from(res.execution_trace.columns, n)
                torch.cuda.synchronize()
                res.close()
                return t_exec

            t_root, t_exec = timed(sharded_run)
            sharded_root = cap_root[0].cpu().numpy().view(np.uint32).tolist()
            exec_all = [None] * world
            dist.all_gather_object(exec_all, t_exec)
            t_ind, t_exec_i = timed(independent_run)
            for _ in range(2):                                    # leave `tree` / `gathered` / cap_root holding the sharded run's commitment again
                step()
            torch.cuda.synchronize()
            multi_e2e = {
                "one_run_row_sharded": {
                    "rows": total_rows, "time_to_root_s": t_root, "end_to_end_rows_per_s_incl_host": total_rows / t_root,
                    "zkir_exec_window_s_per_rank": exec_all, "root": sharded_root,
                    "what": "barrier -> every rank: zkir_exec_window(program, rows [g n, (g+1) n)) = rows [0, g n) executed untraced on the rank's own core, its rows "
                            "traced with H2D + K1 streamed under the interpreter -> main trace -> LDE -> Merkle -> all-gather of the roots + cap -> barrier; max over ranks",
                    "bound": "one run is a sequential chain: its time to root cannot drop below (G - 1) n t_untraced + n t_traced + one GPU's commit step, whatever G is"},
                "independent_runs_one_per_gpu": {
                    "runs": world, "rows_per_run": n, "wall_s": t_ind, "rows_per_s_end_to_end_incl_host": world * n / t_ind, "zkir_exec_s_rank0": t_exec_i,
                    "what": "barrier -> every rank: zkir_exec of its OWN 2^k-cycle run (one interpreter thread per run) -> main trace -> LDE -> Merkle -> barrier: "
                            "G runs need G interpreters, so this is the aggregate that grows with G"}}
        except Exception as e:                                    # the bench line must not depend on this leg
            multi_e2e = {"error": repr(e)}

    # ---- the drop-in entry point itself: zkir_exec = host interpretation + H2D + K1 in one call (VM::new + VM::run, trace left in HBM)
    exec_s = None
    if not dist_mode and k <= 24:
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()
            ts.append(time.perf_counter() - t0)
            assert res.cycles == n
            res.close()
        exec_s = min(ts[1:])                              # steady state (first call: device allocations, cold block pool)

    # ---- the same proof with the host in the loop: independent runs pipelined through interpret -> H2D -> K1 -> prove ----------
    pipelined = pipelined_commit = None
    if commit and not dist_mode and k <= 22 and not args.no_prove:
        from zkir_amd import service
        job = (blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
        service.prove_many([job] * 2, k, producers=1, ctx=ctx, keep_proofs=False)
        service.prove_many([job] * 4, k, producers=2, ctx=ctx, keep_proofs=False)      # second proving thread: its context is created here
        rep = service.prove_many([job] * 48, k, producers=3, ctx=ctx, keep_proofs=False)
        service.prove_many([job] * 2, k, producers=1, ctx=ctx, keep_proofs=False, commit_only=True)
        rc = service.prove_many([job] * 64, k, producers=3, ctx=ctx, keep_proofs=False, commit_only=True)
        pipelined_commit = {"runs": rc.runs, "producer_threads": 3, "ms_per_committed_run": rc.ms_per_run, "rows_per_s_committed_end_to_end": rc.rows_per_s,
                            "note": "host interpretation + H2D + the commit step (trace fill, main trace, LDE, Merkle) of independent runs, host threads overlapped with the "
                                    "GPU: the rate of the bench step with the host and PCIe in the loop"}
        pipelined = {"runs": rep.runs, "producer_threads": 3, "ms_per_proven_run": rep.ms_per_run, "rows_per_s_proven_end_to_end": rep.rows_per_s,
                     "interpret_ms_per_run": rep.interpret_s / rep.runs * 1e3, "upload_ms_per_run": rep.upload_s / rep.runs * 1e3,
                     "proving_threads": 2,
                     "note": "host interpretation + H2D + trace fill + full proof of independent 2^k-row runs: three interpreting threads and two proving threads (own "
                             "context and stream each: one proof's host transcript round trips are covered by the other's kernels) (zkir_amd/service.py)"}

    # ---- parity spot checks outside the timed region (full parity lives in tests/ -m gpu) ----------
    n_chk = min(4096, n)
    got = trace.registers[:, :n_chk].cpu().numpy().view(np.uint64)
    ev_np = shard.reg_events
    for r in (1, 2, 3, 4):
        e = ev_np[ev_np["reg"] == r]
        pos = np.searchsorted(e["vis"], np.arange(n_chk), side="right") - 1
        assert np.array_equal(got[r], e["value"][pos]), "bench parity spot-check failed"
    root = tree[-4:].cpu().numpy().view(np.uint32).tolist() if commit else None
    roots = None
    if commit and dist_mode:
        roots = [x.cpu().numpy().view(np.uint32).tolist() for x in gathered]
        if cap_root[0] is not None:
            root = cap_root[0].cpu().numpy().view(np.uint32).tolist()
    fill_bytes = pl.trace_fill_bytes(ddl)
    n_events, tile_rows = ddl.n_events, ddl.tile_rows

    # ---- the other single-GPU BASELINE configs, outside the timed region (free the configs[1] buffers first) ----
    by_config = None
    if rank == 0 and not dist_mode and commit and not args.no_by_config and k == 20:
        if commit:
            del m, L, tree
            ctx.close()
        del trace, ddl, fill_args
        torch.cuda.empty_cache()
        by_config = {"configs[2]": _config2_fib_2p24(lib, sp), "configs[4]": _config4_sha_2p22(lib, sp),
                     # the N = 1 point of the N > 1 lines: they run 2^23 rows per GPU (configs[3]'s share), this line 2^20 — a SCALE curve is to be read
                     # against THIS rate (a second strided NTT pass per side at 2^23), not against `value`
                     "2^23 rows on one GPU (the per-GPU problem of the N > 1 lines)": _config2_fib_2p24(lib, sp, 23)}
        try:                                           # 2^26 rows need ~185 GB of HBM: reported when the device has them, never fatal
            if torch.cuda.mem_get_info()[1] >= 250 * (1 << 30):
                by_config["configs[3] on one GPU"] = _config2_fib_2p24(lib, sp, 26)
        except Exception as e:
            by_config["configs[3] on one GPU"] = {"error": repr(e)}
            torch.cuda.empty_cache()

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        value = total_rows * args.steps / wall
        kernels = _commit_kernel_table(k, W, fill_bytes, stage_ms, lib, sp)
        dom = max(kernels, key=lambda q: kernels[q]["ms"])
        traffic, traffic_source = _profiled_traffic(dom) if (k == 20 and not dist_mode and commit) else (None, None)
        dom_kernel = {"merkle_leaves": "leaf_hash_kernel", "trace_fill": "trace_fill_kernel", "main_trace": "main_trace_kernel"}.get(dom, dom)
        alu = _alu_roofline(dom_kernel, kernels[dom]["ms"]) if (kernels[dom]["bound"] != "hbm" and k == 20 and not dist_mode) else None
        out = {
            "metric": f"trace rows/sec (2^{k}-cycle fib: execution-trace fill + BabyBear NTT/LDE + Poseidon2 Merkle commitment)"
                      if commit else f"trace rows/sec (2^{k}-cycle fib, execution-trace fill only)",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (Baby Bear, 31-bit modular) / u64 trace words" if commit else "u64", "data": "synthetic",
            "config": {"workload": (f"fib_endless 2^{k} cycles per GPU" + (" = BASELINE configs[1]" if k == 20 and not dist_mode else "")
                                    + (f" = BASELINE configs[3] (2^26-cycle fib row-sharded over 8 GPUs)" if k == 23 and world == 8 else
                                       (f" (configs[3]'s per-GPU share; {world} row shards of one 2^{k + world.bit_length() - 1}-cycle run)" if dist_mode else ""))
                                    + "; v3.4 fib loop of tests/cross_module.rs:145-164, max_cycles halt, VMConfig{enable_execution_trace}; "
                                    f"stages = {'+'.join(s for s, _ in stages)}; blow-up 2, Poseidon2 width 12"),
                       "rows_per_gpu": n, "tile_rows": tile_rows, "reg_events_per_gpu": n_events, "main_trace_width": W if commit else None,
                       "parallelism": f"row-shard x{world}",
                       #  what scales with G and what cannot: said first, not in a nested field
                       "reading_at_n_gt_1": ("`value` = the GPUs' commit throughput on row shards already in HBM (weak scaling: per-GPU work fixed).  END TO END one run is a sequential chain — "
                                             "its time to root is bounded by one host core whatever G is (multi_gpu_end_to_end.one_run_row_sharded.bound) — so the figure that grows with G is "
                                             "the AGGREGATE of G independent runs, one interpreter per GPU: multi_gpu_end_to_end.independent_runs_one_per_gpu.rows_per_s_end_to_end_incl_host") if dist_mode else None},
            # the dominant KERNEL of the step (its own launch bracketed by HIP events inside the timed region; at N = 1 / 2^20 rows: leaf_hash_kernel).  `achieved` / `peak` /
            # `frac` are the HBM figures the contract asks for; `bound` names what the kernel is really bound by, and `alu` prices it against THAT roofline
            "roofline": {"bound": "valu-issue" if kernels[dom]["bound"] != "hbm" else "hbm", "kernel": dom_kernel, "stage": dom, "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": kernels[dom]["frac_of_hbm_peak"], "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_source_freshness": _check_traffic_source_is_fresh(traffic_source) if traffic_source else None,
                         "kernel_ms": kernels[dom]["ms"], "algorithmic_bytes_per_launch": kernels[dom]["bytes"],
                         "alu": alu},
            # `value` is rows/s of a commit over W self-chosen columns: the width-independent figures are per column of 2^20 rows
            "per_column": ({"main_trace_width": W, "rows": n,
                            "lde_us_per_column_per_2p20_rows": stage_ms["lde"] * 1e3 / W * ((1 << 20) / n),
                            "merkle_us_per_column_per_2p20_rows": (stage_ms["merkle_leaves"] + stage_ms["merkle_levels"]) * 1e3 / W * ((1 << 20) / n),
                            "main_trace_us_per_column_per_2p20_rows": stage_ms["main_trace"] * 1e3 / W * ((1 << 20) / n)} if commit else None),
            "roofline_by_stage": kernels,
            "by_config": by_config,
            # rows per second of ONE GPU over its own stages (everything but the all-gather + cap): at N > 1 the shard is 2^23 rows, not the 2^20 of
            # N = 1 (a second strided NTT pass per side), so this — not value(1) — is the single-device rate `value` / N is to be held against
            "per_gpu_local_stage_rows_per_s": n / (sum(v for q, v in stage_ms.items() if q != "allgather_cap") * 1e-3),
            # N > 1: `value` against N x (rank 0's rate over its own stages at the SAME 2^k rows) — what the all-gather + cap and the barrier cost, with the
            # size effect (N = 1 runs 2^20 rows, N > 1 runs 2^23 per GPU) taken out; N = 1: the 2^23-row single-GPU rate the N > 1 lines are to be held against
            "efficiency_vs_same_size_single_gpu": (value / (world * n / (sum(v for q, v in stage_ms.items() if q != "allgather_cap") * 1e-3))) if dist_mode else None,
            "single_gpu_commit_rows_per_s_by_log2_rows": ({str(k): value, **{str(v["rows"].bit_length() - 1): v["commit_rows_per_s"] for v in (by_config or {}).values()
                                                                                 if isinstance(v, dict) and "commit_rows_per_s" in v}} if not dist_mode and commit else None),
            "process_group": ({"backend": backend, "world_size": world, "forced_at_world_1": world == 1,
                               "collectives_executed": ["barrier", "all_gather(int32[4] root, device)" if backend == "nccl" else "all_gather(int32[4] root, host-staged)",
                                                        "all_reduce(f64 MAX)", "all_gather_object", "broadcast_object_list", "gather_object"]} if dist_mode else None),
            "hbm_copy_GBs_measured": hbm_copy_gbs,         # 1 GiB device-to-device copy, read + write bytes / time
            "gpu_ms_per_step_hip_events": gpu_ms_per_step,
            "prove_ms": prove_ms, "prove_ms_mode2": prove_ms_mode2, "prove_ms_100bit": prove_ms_100bit, "prove_stage_ms": prove_stage_ms, "proof_bytes": proof_bytes, "verify_ms_host": verify_ms,
            "stage_ms": stage_ms,
            "prove_stage_roofline": _prove_stage_table(k, W, [prove_stage_ms[q] for q in PROVE_STAGES]) if prove_stage_ms else None,
            "prover": "ZKIR-STARK, AIR v6 (self-defined; 172 logical main-trace columns, 152 committed in default mode, + 40 aux columns / 398 constraints: the semantics of 20 of the 50 opcodes — ADD, ADDI, SUB, SLTU/SGEU/SLT/SGE, SEQ/SNE, CMOV/CMOVZ/CMOVNZ, BEQ/BNE, BLTU/BGEU/BLT/BGE, JAL, JALR — and the control flow of every opcode, + a LogUp lookup argument — instruction ROM and 10-bit ranges, eight range lookups per row; "
                      "boundary states for segment proofs, blow-up 2, 50 queries + 12-bit grinding, Poseidon2-12; proof format v10 carries the program).  `prove_ms` is MODE 0, the default; "
                      "opt-in modes (round 4, `prove_by_mode`): 2 = + the I/O argument (what the run read / wrote is proven), 3 = + the memory argument, the bitwise opcodes and the shifts "
                      "(43 of 50 opcodes, memory consistency; 264 + 96 columns, 636 constraints; the memory witness is made on the device), 4 (round 6, format v12) = 3 + MULH / DIVU / REMU / DIV / REM on operands "
                      "below 2^40 by a chunk relation and on raw 64-bit operands through a verifier-recomputed wide tape (all 50 opcodes carry a statement), the hash syscalls through a tape the verifier recomputes "
                      "(SHA-256 / Keccak / BLAKE3 digests are NOT arithmetised: the verifier hashes the tape's inputs itself), the boundary cell between code and data (288 + 128 columns, 712 constraints)",
            "prove_by_mode": prove_by_mode,
            "pipelined_end_to_end": pipelined, "pipelined_commit_end_to_end": pipelined_commit, "segment_prove": segment_prove,
            "merkle_root": root, "merkle_roots_all_ranks": roots, "allgather_cap_ms": stage_ms.get("allgather_cap"),
            "merkle_root_equals_oracle_golden": _root_vs_golden(k, root) if (commit and not dist_mode) else None,
            "host_interpret_rows_per_s": total_rows / host_s, "host_interpret_first_run_rows_per_s": total_rows / host_first_s,
            "host_interpret_ns_per_instruction": host_s / total_rows * 1e9, "host_interpret_s": host_s,
            # N = 1: the one interpretation of the run.  N > 1: EVERY rank executes its own prefix (rank g: g n rows untraced + n rows traced, on its own
            # core: zkir_interpret_window) — G interpreters per node, no shard transport; `host_interpret_s` is then the slowest rank's (the last one's)
            "host_interpretations_per_node": world, "shard_distribution_s": 0.0,
            "host_window_s_per_rank": per_rank_host_s if dist_mode else None,
            "multi_gpu_end_to_end": multi_e2e,
            "end_to_end_rows_per_s_incl_host": (multi_e2e or {}).get("one_run_row_sharded", {}).get("end_to_end_rows_per_s_incl_host") if dist_mode else None,
            "h2d_upload_s": h2d_s, "h2d_upload_first_in_process_s": h2d_first_s,
            "end_to_end_rows_per_s_incl_host_and_pcie": n / (exec_s + (gpu_ms_per_step - stage_ms["trace_fill"]) * 1e-3) if exec_s else None,
            "zkir_exec_ms": exec_s * 1e3 if exec_s else None,                 # drop-in call: interpret + H2D + trace fill, PCIe-inclusive
            "zkir_exec_rows_per_s": n / exec_s if exec_s else None,
            "host_cpu": _host_cpu(),
        }
        if not args.no_cpu_baseline and not dist_mode:
            out["cpu_baseline"] = _cpu_baseline(blob, k, commit)
            side = out["cpu_baseline"].get("commit_stage_self_defined")
            if side and root is not None:
                assert side["merkle_root"] == root, "bench: the CPU port's commitment root differs from the GPU's"
            c2 = (by_config or {}).get("configs[2]") or {}
            if c2.get("zkir_exec_rows_per_s"):
                # north_star's target, stated on its own terms: the trace path (VM::run -> execution trace, the only stage the reference has) at
                # 2^24 cycles on one GPU against the CPU restatement of the reference in linear mode (faithful mode cannot finish at 2^24)
                l24 = out["cpu_baseline"].get("linear_2p24") or {}
                cpu = l24.get("rows_per_s") or out["cpu_baseline"]["linear_rows_per_s"]
                out["target_10x_at_2p24"] = {"gpu_path": "zkir_exec at 2^24 cycles (host interpretation + H2D + trace fill, 372 B/row left in HBM)",
                                             "gpu_path_rows_per_s": c2["zkir_exec_rows_per_s"], "cpu_linear_rows_per_s": cpu, "cpu_rows": l24.get("rows") or out["cpu_baseline"]["linear_rows"],
                                             "ratio": c2["zkir_exec_rows_per_s"] / cpu,
                                             "what_the_ratio_is": "the product's trace path (an optimised host interpreter writing a 44 B/row delta log, with the GPU expanding it to 372 B/row underneath) "
                                                                  "against the literal single-threaded restatement of VM::run building 372 B rows on the CPU, same size; of the "
                                                                  "GPU path's time ~97 % is the host interpreter and ~3 % K1 — it is an end-to-end ratio of two trace paths, not a kernel speed-up",
                                             "commit_step_ratio_vs_cpu_commit_port": (c2["commit_rows_per_s"] / side["rows_per_s"]) if side else None}
        emit(out)
    if dist_mode:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
