#!/usr/bin/env python3
"""Randomised soak of the prover kernels against the oracle (not collected by pytest; run by hand on a GPU box):

    python tests/soak_gpu_parity.py [seconds=120] [seed=1]

Loops over random shapes and contents — Merkle commitments (random width / leaves, values drawn from edge-heavy
distributions), coset LDEs, and whole proofs of random programs — and compares every output word with the oracle's.
The lazy (unreduced) arithmetic of the kernels is bounded on paper (babybear.h / poseidon2.h); this hunts for a counterexample.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from oracle import api as oracle, stark_api as so
from zkir_amd import pipeline as pl, runtime as rt, spec, stark

import programs

P = so.P
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def edge_heavy(shape):
    """Field elements with a lot of mass at 0, 1, p-1, p-2, 2^31-ish patterns and powers of two."""
    x = rng.integers(0, P, shape).astype(np.uint32)
    kind = rng.integers(0, 8, shape)
    x = np.where(kind == 0, P - 1, x)
    x = np.where(kind == 1, 0, x)
    x = np.where(kind == 2, P - 1 - rng.integers(0, 4, shape), x)
    x = np.where(kind == 3, 1 << rng.integers(0, 31, shape), x) % P
    return x.astype(np.uint32)


t_end = time.time() + budget
n_merkle = n_lde = n_proof = n_wit = n_refused = 0
n_by_mode = [0, 0, 0, 0, 0]
ctxs = {}
while time.time() < t_end:
    what = rng.integers(0, 10)
    if what < 5:                                                    # Merkle
        log_m, width = int(rng.integers(1, 13)), int(rng.integers(1, 100))
        mat = edge_heavy((width, 1 << log_m))
        ctx = ctxs.setdefault(max(log_m - 1, 1), stark.StarkContext(max(log_m - 1, 1)))
        tree = stark.merkle_commit(ctx, stark.to_b8(torch.from_numpy(mat.view(np.int32)).cuda()), mat.shape[0]).cpu().numpy().view(np.uint32)
        _, layers = so.merkle(mat, want_layers=True)
        assert np.array_equal(tree, layers), ("merkle", log_m, width)
        n_merkle += 1
    elif what < 8:                                                  # LDE
        log_n, width = int(rng.integers(1, 15)), int(rng.integers(1, 6))
        mat = edge_heavy((width, 1 << log_n))
        ctx = ctxs.setdefault(log_n, stark.StarkContext(log_n))
        got = stark.from_b8(stark.lde(ctx, stark.to_b8(torch.from_numpy(mat.view(np.int32)).cuda())), mat.shape[0]).cpu().numpy().view(np.uint32)
        for k in range(width):
            assert np.array_equal(got[k], so.lde(mat[k], 1)[1]), ("lde", log_n, k)
        n_lde += 1
    elif what == 8:                                                 # witness kernels on a random program (any length, any halt)
        blob, inputs = programs.random_program(int(rng.integers(0, 1 << 30)), n_instr=int(rng.integers(50, 600)))
        cfg = dict(max_cycles=int(rng.integers(100, 30000)), enable_execution_trace=True, enable_range_checking=bool(rng.integers(0, 2)),
                   enable_deferred_model=bool(rng.integers(0, 2)))
        try:
            want = oracle.run(blob, inputs, **cfg)
        except oracle.OracleError:
            continue
        log = rt.interpret(blob, inputs, rt.VMConfig(**cfg))
        if len(log.mem_events):
            rows_c, offsets, srt = pl.memory_ops(log)
            assert np.array_equal(rows_c.to_numpy(), want.memops) and np.array_equal(srt.to_numpy(), want.sorted_memops), "memops"
            assert np.array_equal(offsets.cpu().numpy().view(np.uint64), want.row_memop_offsets), "memop offsets"
        if len(log.rc_events):
            value, pc, chunks, mult = pl.range_checks(log)
            assert np.array_equal(value.cpu().numpy().view(np.uint64), want.rc_checks["value"]) and np.array_equal(chunks.cpu().numpy().view(np.uint16).T, want.rc_checks["chunks"]), "rc"
        if len(log.norm_events):
            assert np.array_equal(pl.normalization_events(log), want.norm_events), "norm"
        n_wit += 1
        log.close()
    else:                                                           # whole proof of a random program
        log_n = int(rng.integers(3, 11))
        # the proof's MODE: 0 default, 1 deferred model, 2 default + the I/O argument, 3 = 2 + the memory argument (round 4; no hash syscalls there: a run that executes one
        # has no mode-3 proof — the refusal itself is tested in tests/test_gpu_stark.py)
        # (round 6) 4 = 3 + the wide-arithmetic class, the hash tape and the boundary cell: hash syscalls are provable there, the wide opcodes on operands below 2^40
        mode = int(rng.integers(0, 5))
        blob, inputs = programs.random_program(int(rng.integers(0, 1 << 30)), n_instr=200, hashes=mode != 3, wide_safe=mode == 4 and bool(rng.integers(0, 2)))     # (mode 4: half the draws keep the wide opcodes inside the chunk relation's domain, half feed them raw registers — the wide tape)
        cfg = dict(max_cycles=int(rng.integers((1 << log_n) // 2 + 1, (1 << log_n) + 1)), enable_execution_trace=True, enable_deferred_model=mode == 1)
        try:
            res = oracle.run(blob, inputs, **cfg)
        except oracle.OracleError:
            continue
        want_rows = res.rows
        if len(want_rows) == 0:
            continue
        log_n = so.padded_log_n(len(want_rows))                     # any halt: the trace is padded to a power of two
        fri = dict(num_queries=84, pow_bits=16) if rng.integers(0, 4) == 0 else {}                 # (round 5) one proof in four at the second parameter set
        opub = so.public_inputs(len(want_rows), blob, list(inputs), list(res.outputs), (res.halt_kind, res.halt_code), deferred=mode == 1, io_mode=mode == 2, mem_mode=mode == 3, wide_mode=mode == 4, **fri)
        log = rt.interpret(blob, inputs, rt.VMConfig(**cfg))
        ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
        got_rows = tr.rows()
        for name in want_rows.dtype.names:                          # K1 (trace fill) against the oracle's rows, all 372 B
            assert np.array_equal(got_rows[name], want_rows[name]), ("trace", name)
        ctx = ctxs.setdefault(log_n, stark.StarkContext(log_n))
        want = so.prove(want_rows, opub)
        try:
            witness = "host" if (mode >= 3 and rng.integers(0, 2)) else "device"          # mode 3: the memory witness from the device (sort + scan) or from the host replay
            proof = stark.prove(ctx, tr, rt.public_inputs(log, blob, inputs, mode == 1, io_mode=mode == 2, mem_mode=mode == 3, wide_mode=mode == 4, mem_witness=witness, **fri))
        except rt.RuntimeError as e:
            # a run that executes a word that is not the program's (a store into the code segment, a pc outside it) has no proof; the
            # honest GPU prover refuses it — and the proof the oracle's prover emits for the same rows must be one the verifiers reject
            assert e.code == rt.ERR_ARGUMENT and ("code table" in e.message or "2^40" in e.message or "overlaps the code segment" in e.message), e.message
            assert so.verify(want) != 0 and rt.verify(want) == so.verify(want), "refused by the prover but accepted by a verifier"
            n_refused += 1
            log.close()
            continue
        assert np.array_equal(proof, want), ("proof", log_n)
        # COMPLETENESS: an honest execution of ANY program the prover does not refuse satisfies the AIR — every opcode class, every halt, both
        # modes: both verifiers accept (since AIR v4 class "other" is sequential and JALR / the branches are stated: a random program is
        # the test that no opcode advances the pc or writes registers in a way the constraints did not foresee)
        v_o, v_p = so.verify(proof, opub), rt.verify(proof)
        assert v_o == 0 and v_p == 0, ("an honest run was rejected", v_o, v_p, log_n, cfg)
        n_proof += 1
        n_by_mode[mode] += 1
        log.close()
print(f"soak ok: {n_merkle} Merkle trees, {n_lde} LDEs, {n_wit} witness sets, {n_proof} traces + proofs identical to the oracle, "
      f"(modes 0 / 1 / 2 / 3 / 4: {n_by_mode[0]} / {n_by_mode[1]} / {n_by_mode[2]} / {n_by_mode[3]} / {n_by_mode[4]}), "
      f"{n_refused} unprovable runs refused by the GPU prover and rejected by both verifiers, in {budget:.0f} s")
