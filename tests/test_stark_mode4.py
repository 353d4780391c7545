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
C_OM, C_OD, C_ORR, C_GF, C_WE, C_X, C_IWS, C_NB, C_G, C_FH, C_KST, C_E = 284, 285, 286, 287, 291, 300, 306, 307, 167, 175, 181, 182
C_XB, C_XC, C_Y, C_OT, C_RC, C_RC2, C_PIECE = 118, 121, 124, 131, 135, 163, 206       # (C_OT: the column of the class "other", which mode 4 does not have)
M40 = (1 << 40) - 1


def _case(blob, ins=(), **cfg):
    ores = oracle.run(blob, list(ins), enable_execution_trace=True, **cfg)
    pub = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), wide_mode=True)
    return ores, pub


def test_widths():
    assert (so.logical_width(4), so.committed_width(4), so.aux_width(4), so.lib().so_num_constraints_for(4)) == (308, 288, 128, 712)
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


def _signed64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def _ref_wide(op, a, b):
    """execute.rs:101-183 on raw 64-bit registers, in Python integers."""
    O = spec.Opcode
    M64 = (1 << 64) - 1
    if op == O.MULH:
        return ((a * b) >> 40) & M40
    if op == O.DIVU:
        return a // b
    if op == O.REMU:
        return a % b
    sa, sb = _signed64(a), _signed64(b)
    q = abs(sa) // abs(sb) * (1 if (sa < 0) == (sb < 0) else -1)                               # Rust's `/` truncates towards zero; wrapping: i64::MIN / -1 = i64::MIN
    r = sa - q * sb
    return (q if op == O.DIV else r) & M64


def test_wide_operands_above_40_bits_go_through_the_wide_tape():
    """loads_stores ends with DIV / REM / MULH on a register LB sign-extended to 0xFFFF_FFFF_FFFF_FF80 (quirk Q2's test): outside the chunk relation's domain.  Such rows are proven
    through the WIDE TAPE: ot = 1, the proof carries (cycle, rs1, rs2, opcode), the verifier computes the reference's result on the raw 64-bit registers.  The run has a mode-4
    proof; both verifiers accept it; its tape holds exactly the rows with an operand above 2^40."""
    from zkir_amd import runtime as rt, stark
    blob, ins, cfg = pg.off_code("loads_stores")
    ores, pub = _case(blob, ins)
    proof = so.prove(ores.rows, pub)
    assert so.verify(proof, pub) == 0 and rt.verify(proof) == 0
    lay = stark.proof_layout(proof)
    n_tape = int(proof[lay["wide_section"]])
    ops = ores.rows["instruction"] & 0x7F
    M = so.main_trace(ores.rows, pub)
    wide_rows = [i for i in range(len(ores.rows)) if 3 <= int(ops[i]) <= 7 and i + 1 < len(ores.rows)]
    hi = [i for i in wide_rows if int(M[C_XB + 2][i]) or int(M[C_XC + 2][i])]
    assert n_tape == len(hi) >= 3 and [int(proof[lay["wide_section"] + 1 + 8 * k]) for k in range(n_tape)] == hi
    assert all(int(M[C_OT][i]) == 1 for i in hi) and int(M[C_OT].sum()) == len(hi)              # ot sits in the column of the class "other", which mode 4 does not have


def test_wide_tape_semantics_on_raw_64_bit_registers():
    """What the tape route writes is the reference's arithmetic on the RAW registers (execute.rs:101-183; quirks Q2 / Q3): signed division truncating towards zero with the
    remainder's sign following the dividend, i64::MIN / -1 wrapping to i64::MIN with remainder 0, unsigned division of values above 2^63, MULH = bits 40..79 of the 128-bit
    product — checked against Python integers on a grid of sign-extended, 64-bit and mixed operands that the VM executes (the inputs arrive by READ)."""
    from zkir_amd import runtime as rt
    O, E = spec.Opcode, spec.encode
    vals = [0xFFFFFFFFFFFFFF80, 0x8000000000000000, 0xFFFFFFFFFFFFFFFF, 0x7FFFFFFFFFFFFFFF, 0x0000010000000000, 0x00000000000000FF, 0x123456789ABCDEF0, 3]
    regs = [1, 2, 3, 4, 9, 14, 15, 7]
    code = [pg.A(5, 0, 1)]
    for k, r in enumerate(regs):
        code += [pg.A(10, 0, 1), pg.EC, E(O.CMOV, r, 10, 5)]                                      # READ leaves the raw 64 bits in r10; CMOV copies them (ADDI would cut them to 40)
    pairs = [(a, b) for a in range(len(vals)) for b in range(len(vals))]
    body = []
    for a, b in pairs:
        for op in (O.MULH, O.DIVU, O.REMU, O.DIV, O.REM):
            body.append(E(op, 8, regs[a], regs[b]))
    blob = pg._p(code + body + [pg.EB])
    ores, pub = _case(blob, vals)
    rows = ores.rows
    at = len(code)
    for a, b in pairs:
        for op in (O.MULH, O.DIVU, O.REMU, O.DIV, O.REM):
            got = int(rows["registers"][at + 1][8])
            assert got == _ref_wide(op, vals[a], vals[b]), (hex(vals[a]), hex(vals[b]), op, hex(got))
            at += 1
    proof = so.prove(rows, pub)
    assert so.verify(proof, pub) == 0 and rt.verify(proof) == 0
    M, cl = so.main_trace(rows, pub), so.mem_cells(rows, pub)
    assert so.failing_constraints(M, pub, cl)[0] == 0
    assert int(M[C_OT].sum()) >= 5 * (len(pairs) - 4)                                            # nearly every row of the grid has an operand above 2^40 (only 0xFF / 3 pairs do not)


def test_a_forged_wide_tape_is_rejected():
    """The tape is public and the verifier recomputes it: a record with another operand, another opcode, a missing or an extra record, a zero divisor, records out of order — each is
    rejected by both verifiers (57: malformed; 10: the row's lookup finds nothing; the Merkle / sum checks otherwise).  A row that claims the tape without a record, and an
    in-domain row forced through the tape WITH its record (sound: the verifier computes it), behave as stated."""
    from zkir_amd import runtime as rt, stark
    blob, ins, cfg = pg.off_code("loads_stores")
    ores, pub = _case(blob, ins)
    proof = so.prove(ores.rows, pub)
    lay = stark.proof_layout(proof)
    w0 = lay["wide_section"]
    n = int(proof[w0])
    assert n >= 3

    def both(t):
        a, b = so.verify(t), rt.verify(t)
        assert a == b, (a, b)
        return a
    t = proof.copy(); t[w0 + 1 + 1] ^= 1                                                        # another rs1 in the first record
    assert both(t) != 0
    t = proof.copy(); t[w0 + 1 + 7] = 3 if int(t[w0 + 1 + 7]) != 3 else 4                       # another opcode
    assert both(t) != 0
    t = proof.copy(); t[w0 + 1 + 7] = 8
    assert both(t) == 57                                                                       # not a wide opcode
    t = proof.copy(); t[w0 + 1 + 4] = t[w0 + 1 + 5] = t[w0 + 1 + 6] = 0; t[w0 + 1 + 7] = 4
    assert both(t) == 57                                                                       # a division by zero never is a row
    t = proof.copy(); t[w0 + 1], t[w0 + 9] = proof[w0 + 9], proof[w0 + 1]
    assert both(t) == 57                                                                       # records out of order
    t = proof.copy(); t[w0 + 1 + 3] = 1 << 24
    assert both(t) == 57                                                                       # a limb out of range
    t = np.concatenate([proof[:w0], [n - 1], proof[w0 + 1:w0 + 1 + 8 * (n - 1)], proof[w0 + 1 + 8 * n:]]).astype(np.uint32)
    assert both(t) != 0                                                                        # a record missing
    t = np.concatenate([proof[:w0], [n + 1], proof[w0 + 1:w0 + 1 + 8 * n], [int(proof[w0 + 1 + 8 * (n - 1)]) + 1, 5, 0, 1, 7, 0, 0, 4], proof[w0 + 1 + 8 * n:]]).astype(np.uint32)
    assert both(t) != 0                                                                        # a record no row looks up
    # the rows themselves: the matrix of an honest prover with one tape row's result changed has no valid proof (its lookup finds the verifier's result, not the forged one)
    M, cl = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    i = int(proof[w0 + 1])
    M2 = M.copy(); M2[C_Y][i] = (int(M2[C_Y][i]) + 1) % so.P
    fc = so.failing_constraints(M2, pub, cl)
    assert fc[0] >= 1
    # an IN-DOMAIN wide row may take the tape too (any prover may; the honest one does not): clear its chunk columns, set ot — every constraint still holds
    blob2 = spec.wide_loop_program().to_bytes()
    o2, p2 = _case(blob2, max_cycles=200)
    M3, c3 = so.main_trace(o2.rows, p2), so.mem_cells(o2.rows, p2)
    ops = o2.rows["instruction"] & 0x7F
    j = int(np.nonzero((ops >= 3) & (ops <= 7))[0][2])
    for c in list(range(284, 306)) + list(range(C_PIECE, C_PIECE + 9)) + list(range(C_RC, C_RC + 4)) + list(range(C_RC2, C_RC2 + 4)):
        M3[c][j] = 0
    M3[C_OT][j] = 1
    assert so.failing_constraints(M3, p2, c3)[0] == 0


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
    kwa = M[C_OM] + M[C_OD] + M[C_ORR]
    assert np.array_equal(kwa[:nr] != 0, (ops >= 3) & (ops <= 7) & (np.arange(nr) < nr - 1))
    assert np.array_equal(M[C_OM][rows] != 0, ops[rows] == 3) and np.array_equal(M[C_G][rows] != 0, ops[rows] >= 6)       # DIV / REM: the word's variant bit
    assert [int(M[C_WE + k][rows].max()) for k in range(8)] == [1] * 8            # every carry bit but the last occurs (c5 = 2048 would need both operands all ones AND c4 maximal)

    # the forgeries on a SMALL grid (the same operand pairs; a quarter of the rows: each forgery is a constraint sweep and a proof)
    blob, ins, _ = pg.wide_grid([0, 1, 0xF0F0A5C3E1, 0x0312345678, 1024, 0xFFFFFFFFFF])
    ores, pub = _case(blob)
    M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    regs, words = ores.rows["registers"], ores.rows["instruction"]
    ops = words & 0x7F
    nr = len(ops)
    rows = np.nonzero((ops >= 3) & (ops <= 7) & (np.arange(nr) < nr - 1))[0]
    assert so.failing_constraints(M, pub, cells)[0] == 0

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
    assert bad(lambda F: (F.__setitem__((C_OM, k), 1), F.__setitem__((C_OD, k), 0)))                          # a DIV word run as MULH
    k0 = int(np.nonzero(ops == 0x08)[0][0])
    assert bad(lambda F: F.__setitem__((C_OD, k0), 1))                                                       # an ADDI run as a wide row
    assert bad(lambda F: (F.__setitem__((C_OD, i), 1), F.__setitem__((C_ORR, i), 1)))                        # two kinds at once
    assert bad(lambda F: F.__setitem__((C_GF, k0), 5))                                                       # F1's gated copies off the wide rows
    assert bad(lambda F: F.__setitem__((C_X + 3, k0), 1024))                                                 # an extra range slot outside the table


def _pub_c(p):
    from zkir_amd import runtime as rt
    out = rt.PublicInputsC(p.n_real, p.entry, p.deferred, 0)
    out.program_digest[:] = list(p.prog); out.io_digest[:] = list(p.io)
    return out


@pytest.mark.parametrize("name", ["wide_grid", "alu_all", "timestamps", "random3", "loads_stores"])
def test_product_verifier_agrees_with_the_oracle(name):
    """zkir_verify (verify.cpp + air.h: the product's own constraint list) on the oracle's mode-4 proofs: same verdict on honest proofs — with rows that go through
    the wide tape (loads_stores: DIV / REM / MULH on a sign-extended register; random3: divisions of raw 64-bit values) — and on tampered copies; and on forged wide rows the
    same failing check."""
    from zkir_amd import runtime as rt
    if name.startswith("random"):
        blob, ins = pg.random_program(int(name[6:]), hashes=False); cfg = {}
    else:
        blob, ins, cfg = pg.off_code(name)
    ores, pub = _case(blob, ins, **{k: v for k, v in cfg.items() if k == "max_cycles"})
    pr = so.prove(ores.rows, pub)
    assert so.verify(pr, pub) == 0 and rt.verify(pr) == 0 and rt.verify(pr, _pub_c(pub)) == 0
    for pos in (8, 30, 160, len(pr) // 3, len(pr) - 1):
        t = pr.copy(); t[pos] = (int(t[pos]) + 1) % so.P
        assert so.verify(t) != 0 and rt.verify(t) == so.verify(t), pos
    if name == "wide_grid":
        M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
        i = int(np.nonzero(M[C_OM] + M[C_OD] + M[C_ORR])[0][100])
        F = M.copy(); F[C_PIECE + 5, i] = (int(F[C_PIECE + 5, i]) + 1) % 1024
        forged = so.prove_matrix_mem(F, pub, cells)
        assert so.verify(forged, None) == rt.verify(forged) == 10


# ---- (b) hash syscalls as a tape -------------------------------------------------------------------------------------------------------------------------------
def _hash_case(name):
    if name == "sha_chain":
        blob, ins, cfg = spec.sha256_chain_program().to_bytes(), [], {"max_cycles": 700}
    else:
        blob, ins, cfg = getattr(pg, name)()
        cfg = {k: v for k, v in cfg.items() if k == "max_cycles"}
    ores, pub = _case(blob, ins, **cfg)
    return blob, ins, ores, pub


HASH_PROGRAMS = ["sha_chain", "sha256_hello", "hashes_all", "blake3_multi_chunk"]


@pytest.mark.parametrize("name", HASH_PROGRAMS)
def test_runs_with_hash_syscalls_have_a_proof(name):
    """configs[4]'s program (the SHA-256 hash chain) and the reference's hash-syscall tests: the proof carries one record per call, both verifiers recompute every digest
    and accept; the same run has NO mode-3 proof (check 10: "no hash syscall" is a constraint there)."""
    from zkir_amd import runtime as rt
    blob, ins, ores, pub = _hash_case(name)
    hs = so.hash_section(ores.rows, pub)
    assert hs[0] >= 1
    proof = so.prove(ores.rows, pub)
    assert so.verify(proof, pub) == 0 and rt.verify(proof) == 0 and rt.verify(proof, _pub_c(pub)) == 0
    so.set_hash_calls(hs)
    try:
        assert so.failing_constraints(so.main_trace(ores.rows, pub), pub, so.mem_cells(ores.rows, pub))[0] == 0
    finally:
        so.set_hash_calls(None)
    pub3 = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), mem_mode=True)
    assert so.verify(so.prove(ores.rows, pub3), pub3) in (10, 55)      # (55: the chain program reads its seed from the cell the last code word shares with the data — mode 4 admits that cell)


def test_a_forged_hash_tape_is_rejected():
    """The tape is the prover's claim; what makes it binding: (1) every digest is the VERIFIER's own computation over the message the record's cells hold — a proof whose memory
    shows another digest does not balance; (2) the ECALL row looks its (cycle, pointers, length, kind) up in the tape — a record with another pointer, another length, a missing
    or an extra record breaks the lookup; (3) the cells' previous-access times are checked in the clear (a record that reads a cell "from the future": check 56).  Both
    verifiers, same verdicts."""
    from zkir_amd import runtime as rt
    blob, ins, ores, pub = _hash_case("sha_chain")
    M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    hs = so.hash_section(ores.rows, pub)
    n_calls = int(hs[0])
    assert n_calls > 50

    def verdicts(section):
        so.set_hash_calls(section)
        try:
            pr = so.prove_matrix_mem(M, pub, cells)
        finally:
            so.set_hash_calls(None)
        a, b = so.verify(pr, None), rt.verify(pr)
        assert a == b, (a, b)
        return a
    assert verdicts(hs) == 0                                                      # the honest tape through the same path
    rec = 1 + 3 * (8 + 5 * int(hs[1 + 7]))                                        # the fourth record (all records of the chain have the same shape: 8 + 5 x 8 words)
    assert int(hs[rec + 6]) == 3 and int(hs[rec + 7]) == 8
    t = hs.copy(); t[rec + 8 + 5 * 1 + 1] ^= 1                                    # one bit of the MESSAGE in the record: the verifier hashes another message — the digest in memory is not its digest
    assert verdicts(t) == 10
    t = hs.copy(); t[rec + 3] = int(t[rec + 3]) - 1                               # a shorter input length: another tuple than the row's (and another digest)
    assert verdicts(t) == 10
    t = hs.copy(); t[rec + 6] = 5                                                 # the call recorded as Keccak-256
    assert verdicts(t) == 10
    t = hs.copy(); t[rec + 8] = int(t[rec]) + 1                                   # a cell "last accessed" after the call: the Blum condition, checked in the clear
    assert verdicts(t) == 56
    t = hs.copy(); t[rec + 8] = int(t[rec + 8]) - 1                               # .. or at another earlier time than it was: the memory check does not balance
    assert verdicts(t) == 10
    drop = np.concatenate([[n_calls - 1], hs[1:rec], hs[rec + 48:]]).astype(np.uint32)      # a record missing: its row's lookup finds nothing
    assert verdicts(drop) == 10
    t = hs.copy(); t[rec] = int(t[rec - 48])                                      # records out of cycle order
    assert verdicts(t) == 56


# ---- (c) the boundary cell ---------------------------------------------------------------------------------------------------------------------------------------
def _boundary_program(store_low: bool, odd: bool = True):
    """Code of an ODD number of words (code_size % 8 == 4) followed by data: the last code word and the first four data bytes share the cell at 0x1000 + code_size - 4.
    The program loads the first data word (LW at the boundary cell's upper half), stores into the upper half (SW), loads the code word beside it (LW: the lower half) — and, if
    asked, stores into the LOWER half, i.e. over its own last instruction."""
    A, E, O = pg.A, spec.encode, spec.Opcode
    n_words = 9 if odd else 10
    data_at = 0x1000 + 4 * n_words
    code = [A(5, 0, data_at), E(O.LW, 1, 5, imm=0), A(2, 1, 7), E(O.SW, rs1=5, rs2=2, imm=0), E(O.LW, 3, 5, imm=-4), E(O.LBU, 4, 5, imm=1)]
    code += [E(O.SW, rs1=5, rs2=2, imm=-4)] if store_low else [A(6, 0, 1)]
    code += [pg.EB] + [A(7, 0, 2)] * (n_words - len(code) - 1)                  # (the words behind the EBREAK are never executed: the store may land on the last one)
    assert len(code) == n_words
    return pg._p(code, data=bytes([0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88]))


def test_the_boundary_cell_between_code_and_data():
    """with code_size % 8 == 4 the reference's first data word (vm.rs:163-168) shares a cell with the last code word; mode 3 refuses every cell that overlaps
    the code (check 55), so such a program had no proof.  Mode 4 admits that ONE cell and states that no store writes its low half: loads of either half and stores to the data
    half are proven; a store over the last instruction has no proof (the constraint I_BC fails: 10) — and cells INSIDE the code stay refused (55)."""
    from zkir_amd import runtime as rt
    blob = _boundary_program(store_low=False)
    ores, pub = _case(blob)
    assert int(ores.rows["registers"][-1][1]) == 0x44332211 and int(ores.rows["registers"][-1][4]) == 0x22
    cells = so.mem_cells(ores.rows, pub)
    assert len(cells) == 1 and int(cells[0][0]) == 0x1000 + 4 * 9 - 4                          # the one touched cell IS the boundary cell
    proof = so.prove(ores.rows, pub)
    assert so.verify(proof, pub) == 0 and rt.verify(proof) == 0
    pub3 = so.public_inputs(len(ores.rows), blob, [], list(ores.outputs), (ores.halt_kind, ores.halt_code), mem_mode=True)
    assert so.verify(so.prove(ores.rows, pub3), pub3) == 55                                    # mode 3: refused, as before
    bad = _boundary_program(store_low=True)
    ores, pub = _case(bad)
    pr = so.prove(ores.rows, pub)
    assert so.verify(pr, pub) == 10 and rt.verify(pr) == 10                                    # a store over the last code word
    M, cl = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    fc = so.failing_constraints(M, pub, cl)
    assert fc[0] == 1 and int(fc[1][0][0]) == 702                                              # exactly I_BC + 1: tl (kst - nb) on the storing row
    even = _boundary_program(store_low=False, odd=False)                                      # an even word count: the cell at data_at - 4 .. is all code + the access at -4 touches a code-only cell
    ores, pub = _case(even)
    assert so.verify(so.prove(ores.rows, pub), pub) == 55


@pytest.mark.parametrize("seed", [4, 8, 12])
def test_random_programs_on_raw_registers_are_accepted(seed):
    """The same on programs whose wide opcodes read whatever the registers hold (sign-extended loads, 64-bit inputs, LD results): rows inside the chunk relation's domain and
    rows through the wide tape side by side, with hash syscalls between them."""
    from zkir_amd import runtime as rt
    blob, ins = pg.random_program(seed, n_instr=200, hashes=True)
    ores = oracle.run(blob, ins, max_cycles=600, enable_execution_trace=True)
    pub = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), wide_mode=True)
    proof = so.prove(ores.rows, pub)
    assert so.verify(proof, pub) == 0 and rt.verify(proof) == 0


@pytest.mark.parametrize("seed", [0, 3, 7, 11])
def test_random_programs_with_wide_ops_and_hash_calls_are_accepted(seed):
    """COMPLETENESS of mode 4 on programs nobody designed: 200 random instructions — every opcode class, loads and stores of every width, the five wide opcodes on operands
    cut below 2^40 (zero dividends, equal operands, r0 among them), SHA-256 / Keccak / BLAKE3 syscalls over stored bytes — prove, and both verifiers accept.  (The GPU prover
    runs the same draw, byte-compared with this oracle, in tests/soak_gpu_parity.py.)"""
    from zkir_amd import runtime as rt
    blob, ins = pg.random_program(seed, n_instr=200, hashes=True, wide_safe=True)
    ores = oracle.run(blob, ins, max_cycles=600, enable_execution_trace=True)
    op = ores.rows["instruction"] & 0x7F
    wide = np.isin(op, [int(o) for o in (spec.Opcode.MULH, spec.Opcode.DIVU, spec.Opcode.REMU, spec.Opcode.DIV, spec.Opcode.REM)])
    assert wide.sum() >= 1                                                                     # the draw does exercise the class
    pub = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), wide_mode=True)
    proof = so.prove(ores.rows, pub)
    assert so.verify(proof, pub) == 0 and rt.verify(proof) == 0


def test_mutated_mode4_proofs_are_rejected_by_both_verifiers_alike():
    """Robustness of the two verifiers on the format's newest sections: 600 single-word mutations of a mode-4 proof that carries touched cells and a hash tape — a third of them
    inside the memory / hash sections, with values drawn from the edges (0, 1, 2^16, 2^20, p - 1, 2^32 - 1, the word + 1) — are all rejected, with the same code by
    `zkir_verify` and by the oracle's verifier, and without either of them reading outside the proof (a forged count cannot make the parser allocate beyond what the proof holds)."""
    from zkir_amd import runtime as rt, stark
    blob, ins = pg.random_program(3, n_instr=200, hashes=True, wide_safe=True)
    ores = oracle.run(blob, ins, max_cycles=600, enable_execution_trace=True)
    pub = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), wide_mode=True)
    proof = so.prove(ores.rows, pub)
    assert so.verify(proof) == 0 and rt.verify(proof) == 0
    lay = stark.proof_layout(proof)
    assert lay["mode"] == 4 and lay["rom_mult"] - lay["hash_section"] > 100                  # the tape is there
    rng = np.random.default_rng(2026)
    codes = {}
    for trial in range(600):
        where = trial % 3
        pos = int(rng.integers(lay["mem_section"], lay["rom_mult"])) if where == 0 else int(rng.integers(0, lay["trace_root"])) if where == 1 else int(rng.integers(0, len(proof)))
        t = proof.copy()
        old = int(t[pos])
        new = [0, 1, 1 << 16, 1 << 20, so.P - 1, 0xFFFFFFFF, (old + 1) % so.P, int(rng.integers(0, so.P))][int(rng.integers(0, 8))]
        if new == old:
            continue
        t[pos] = new
        a, b = so.verify(t), rt.verify(t)
        assert a != 0 and b != 0, ("a mutated proof was accepted", pos, old, new, a, b)
        assert a == b, ("the verifiers disagree", pos, old, new, a, b)
        codes[a] = codes.get(a, 0) + 1
    assert len(codes) >= 4                                                                     # format, section, constraint and Merkle / FRI rejections were all exercised
    # truncations: every prefix length around the section boundaries is refused by both, alike
    for cut in (lay["mem_section"] + 1, lay["hash_section"], lay["hash_section"] + 1, lay["hash_section"] + 9, lay["rom_mult"] - 1, len(proof) - 1):
        a, b = so.verify(proof[:cut]), rt.verify(proof[:cut])
        assert a != 0 and a == b, (cut, a, b)
