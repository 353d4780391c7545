"""ZKIR-STARK MODE 4 (round 6, proof format v12): mode 3 with the WIDE-ARITHMETIC class — MULH DIVU REMU DIV REM (opcodes 3..7, execute.rs:101-183) on operands below 2^40,
stated as ONE relation F1 F2 + ADD = LO + 2^40 HI over 10-bit chunks (oracle/stark_oracle.cpp "MODE 4").  The reference computes the five on the raw 64-bit
registers (quirks Q2, Q3); below 2^40 an i64 is non-negative, so DIV = DIVU, REM = REMU and MULH is the 80-bit product's upper half — a run that feeds them a wider register
(a sign-extended LB / LH result, an LD, an input) has NO mode-4 proof.  CPU tests of the oracle's statement; the product's host-side pieces are compared with it in
tests/test_abi.py, the GPU prover in tests/test_gpu_stark.py.  PARITY UNPINNED (the reference has no prover)."""
import numpy as np
import pytest

import programs as pg
from oracle import api as oracle, stark_api as so
from zkir_amd import spec

C_LIMB, C_Y, C_RC, C_RC2, C_PIECE = 9, 124, 135, 163, 206
C_KWA, C_OM, C_OD, C_ORR, C_SG, C_GF, C_WE, C_X = 284, 285, 286, 287, 288, 289, 293, 302
M40 = (1 << 40) - 1


def _case(blob, ins=(), **cfg):
    ores = oracle.run(blob, list(ins), enable_execution_trace=True, **cfg)
    pub = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), wide_mode=True)
    return ores, pub


def test_widths():
    assert (so.logical_width(4), so.committed_width(4), so.aux_width(4), so.lib().so_num_constraints_for(4)) == (308, 288, 120, 704)
    assert (so.logical_width(3), so.committed_width(3), so.aux_width(3), so.lib().so_num_constraints_for(3)) == (284, 264, 96, 636)      # mode 3 untouched


@pytest.mark.parametrize("name", ["wide_grid", "alu_all", "timestamps", "mul_grid", "echo5", "fib30", "jumps_and_links"])
def test_honest_runs_are_accepted(name):
    """Runs whose wide opcodes see 40-bit operands — and runs with none of them: everything mode 3 states is stated here unchanged — have a proof; no constraint is violated."""
    blob, ins, cfg = pg.off_code(name)
    ores, pub = _case(blob, ins, **{k: v for k, v in cfg.items() if k == "max_cycles"})
    proof = so.prove(ores.rows, pub)
    assert proof[1] == 12 and proof[9] == 4 and proof[3] == 288
    assert so.verify(proof, pub) == 0
    assert so.failing_constraints(so.main_trace(ores.rows, pub), pub, so.mem_cells(ores.rows, pub))[0] == 0
    assert so.verify_segment(proof, pub)[0] == 2                                  # the memory check spans the whole run: never a segment


def test_the_endless_wide_loop_at_a_ragged_size():
    blob = spec.wide_loop_program().to_bytes()
    ores, pub = _case(blob, max_cycles=777)
    ops = ores.rows["instruction"] & 0x7F
    assert int(((ops >= 3) & (ops <= 7)).sum()) > 240 and int(ores.rows["registers"].max()) <= M40
    assert so.verify(so.prove(ores.rows, pub), pub) == 0


def test_wide_operands_above_40_bits_have_no_proof():
    """loads_stores ends with DIV / REM / MULH on a register LB sign-extended to 0xFFFF_FFFF_FFFF_FFFF (quirk Q2's test): outside the AIR's domain — I_WA_TOP fails (check 10);
    the same run in mode 3, where the five opcodes are class "other", has a proof."""
    blob, ins, cfg = pg.off_code("loads_stores")
    ores, pub = _case(blob, ins)
    assert so.verify(so.prove(ores.rows, pub), pub) == 10
    pub3 = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), mem_mode=True)
    assert so.verify(so.prove(ores.rows, pub3), pub3) == 0


def test_wide_arithmetic_is_constrained():
    """The VM's results on the grid are floor(a b / 2^40), floor(a / b), a mod b; honest rows satisfy every constraint with the largest carries occurring; a quotient off by one
    (with the remainder moved to match: it leaves [0, b)), a forged carry, a carry "bit" of 2, a remainder equal to the divisor, a REMU word run as a DIVU, a DIV word run as
    MULH, an ADDI run as a wide row, gf off the wide rows: each is rejected at the constraint check by the oracle's verifier."""
    blob, ins, _ = pg.wide_grid()
    ores, pub = _case(blob)
    M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    regs, words = ores.rows["registers"], ores.rows["instruction"]
    ops = words & 0x7F
    nr = len(ops)
    rows = np.nonzero((ops >= 3) & (ops <= 7) & (np.arange(nr) < nr - 1))[0]
    assert len(rows) > 1000
    for i in rows:
        w = int(words[i]); rd = (w >> 7) & 15; a = int(regs[i][(w >> 11) & 15]); b = int(regs[i][(w >> 15) & 15]); op = int(ops[i])
        assert a <= M40 and b <= M40
        want = ((a * b) >> 40) & M40 if op == 3 else a // b if op in (4, 6) else a % b
        if rd:
            assert int(regs[i + 1][rd]) == want
    assert so.failing_constraints(M, pub, cells)[0] == 0
    assert np.array_equal(M[C_KWA][:nr] != 0, (ops >= 3) & (ops <= 7) & (np.arange(nr) < nr - 1))
    assert np.array_equal(M[C_OM][rows] != 0, ops[rows] == 3) and np.array_equal(M[C_SG][rows] != 0, ops[rows] >= 6)
    assert [int(M[C_WE + k][rows].max()) for k in range(8)] == [1] * 8            # every carry bit but the last occurs (c5 = 2048 would need both operands all ones AND c4 maximal)

    def bad(edit):
        F = M.copy(); edit(F)
        return so.failing_constraints(F, pub, cells)[0] > 0 and so.verify(so.prove_matrix_mem(F, pub, cells), None) == 10
    i = int([r for r in rows if int(ops[r]) == 4 and int(regs[r][1]) == 0xF0F0A5C3E1 and int(regs[r][2]) == 0x0312345678][0])       # DIVU r5 = r1 / r2

    def quotient_off_by_one(F):                                                  # q + 1 with r - b: y, F1's chunk, gf and the register that follows agree — the remainder leaves its range
        F[C_Y, i] = int(F[C_Y, i]) + 1; F[C_RC2, i] = int(F[C_RC2, i]) + 1; F[C_GF, i] = int(F[C_GF, i]) + 1; F[C_LIMB + 3 * 5, i + 1:i + 2] = F[C_Y, i]
    assert bad(quotient_off_by_one)
    assert bad(lambda F: F.__setitem__((C_PIECE + 5, i), (int(F[C_PIECE + 5, i]) + 1) % 1024))                # a forged carry
    assert bad(lambda F: F.__setitem__((C_WE, i), 2))                                                        # a carry "bit" of 2
    assert bad(lambda F: F.__setitem__((C_GF + 2, i), (int(F[C_GF + 2, i]) + 1) % 1024))                      # gf_2 is not kwa F1_2
    assert bad(lambda F: F.__setitem__((C_PIECE + 1, i), (int(F[C_PIECE + 1, i]) + 1) % 1024))                # a chunk that is not rs2's
    j = int([r for r in rows if int(ops[r]) == 5 and int(regs[r][1]) == 0xF0F0A5C3E1 and int(regs[r][2]) == 1024][0])                # REMU by 1024

    def remainder_equals_divisor(F):                                             # q - 1, r + b: the product equation still holds, d = b - r - 1 does not exist
        F[C_RC2, j] = int(F[C_RC2, j]) - 1; F[C_GF, j] = int(F[C_GF, j]) - 1
        F[C_PIECE + 8, j] = int(F[C_PIECE + 8, j]) + 1; F[C_Y, j] = int(F[C_Y, j]) + 1024; F[C_LIMB + 3 * 6, j + 1:j + 2] = F[C_Y, j]
    assert bad(remainder_equals_divisor)
    assert bad(lambda F: (F.__setitem__((C_OD, j), 1), F.__setitem__((C_ORR, j), 0)))                        # a REMU word run as a DIVU
    k = int([r for r in rows if int(ops[r]) == 6][3])
    assert bad(lambda F: (F.__setitem__((C_OM, k), 1), F.__setitem__((C_OD, k), 0), F.__setitem__((C_SG, k), 0)))       # a DIV word run as MULH
    k0 = int(np.nonzero(ops == 0x08)[0][0])
    assert bad(lambda F: F.__setitem__((C_KWA, k0), 1))                                                      # an ADDI run as a wide row
    assert bad(lambda F: F.__setitem__((C_GF, k0), 5))                                                       # F1's gated copies off the wide rows
    assert bad(lambda F: F.__setitem__((C_X + 3, k0), 1024))                                                 # an extra range slot outside the table


def _pub_c(p):
    from zkir_amd import runtime as rt
    out = rt.PublicInputsC(p.n_real, p.entry, p.deferred, 0)
    out.program_digest[:] = list(p.prog); out.io_digest[:] = list(p.io)
    return out


@pytest.mark.parametrize("name", ["wide_grid", "alu_all", "timestamps", "random3", "loads_stores"])
def test_product_verifier_agrees_with_the_oracle(name):
    """zkir_verify (verify.cpp + air.h: the product's own constraint list) on the oracle's mode-4 proofs: same verdict on honest proofs, on a proof outside the domain
    (loads_stores: check 10) and on tampered copies; and on forged wide rows the same failing check."""
    from zkir_amd import runtime as rt
    if name.startswith("random"):
        blob, ins = pg.random_program(int(name[6:]), hashes=False); cfg = {}
    else:
        blob, ins, cfg = pg.off_code(name)
    ores, pub = _case(blob, ins, **{k: v for k, v in cfg.items() if k == "max_cycles"})
    pr = so.prove(ores.rows, pub)
    want = 10 if name == "loads_stores" else 0
    assert so.verify(pr, pub) == want and rt.verify(pr) == want and rt.verify(pr, _pub_c(pub)) == want
    if want:
        return
    for pos in (8, 30, 160, len(pr) // 3, len(pr) - 1):
        t = pr.copy(); t[pos] = (int(t[pos]) + 1) % so.P
        assert so.verify(t) != 0 and rt.verify(t) == so.verify(t), pos
    if name == "wide_grid":
        M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
        i = int(np.nonzero(M[C_KWA])[0][100])
        F = M.copy(); F[C_PIECE + 5, i] = (int(F[C_PIECE + 5, i]) + 1) % 1024
        forged = so.prove_matrix_mem(F, pub, cells)
        assert so.verify(forged, None) == rt.verify(forged) == 10
