"""ZKIR-STARK MODE 3 (round 4): the default VM mode with the I/O argument AND the memory argument — the ten loads and stores (execute.rs:477-575) are constrained and every
access is tied to a consistent memory by an offline memory check over aligned 8-byte cells — AND the six bitwise opcodes (execute.rs:199-282), nibble by nibble.  CPU tests of the ORACLE (prover + verifier); the product's
verifier and main-trace code are compared with it in tests/test_abi.py, the GPU prover in tests/test_gpu_stark.py.  PARITY UNPINNED (the reference has no prover)."""
import numpy as np
import pytest

import programs as pg
from oracle import api as oracle, stark_api as so
from zkir_amd import spec

A = pg.A
C_CYCLE, C_LIMB, C_Y, C_RC2, C_OB, C_TOLD, C_PIECE = 0, 9, 124, 163, 197, 205, 206
I_SUM = 394                                            # the four running-sum constraints of the base list: they fail on the wrap-around row exactly when the LogUp sums differ


def _case(blob, ins=(), **cfg):
    ores = oracle.run(blob, list(ins), enable_execution_trace=True, **cfg)
    pub = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), mem_mode=True)
    return ores, pub


def test_widths():
    assert (so.logical_width(3), so.committed_width(3), so.aux_width(3), so.lib().so_num_constraints_for(3)) == (284, 264, 96, 636)
    assert (so.logical_width(2), so.committed_width(2), so.aux_width(2), so.lib().so_num_constraints_for(2)) == (180, 160, 48, 430)      # mode 2 untouched


@pytest.mark.parametrize("name", ["mem_sw_lw", "timestamps", "loads_stores", "alu_all", "mul_grid", "echo5", "fib30", "rc_doubling", "jumps_and_links"])
def test_honest_runs_are_accepted(name):
    """Every width (LB LBU LH LHU LW LD / SB SH SW SD), sign extension, loads of untouched cells and of the program image; no constraint is violated on any row.
    (mem_sw_lw, timestamps, rc_doubling: the reference's tests with their data moved off the code segment — pg.off_code; as written they store at 0x1000: below.)"""
    blob, ins, cfg = pg.off_code(name)
    ores, pub = _case(blob, ins, **{k: v for k, v in cfg.items() if k == "max_cycles"})
    proof = so.prove(ores.rows, pub)
    assert proof[9] == 3 and proof[3] == 264
    assert so.verify(proof, pub) == 0
    assert so.failing_constraints(so.main_trace(ores.rows, pub), pub, so.mem_cells(ores.rows, pub))[0] == 0
    assert so.verify_segment(proof, pub)[0] == 2                                  # the memory check spans the whole run: never a segment


@pytest.mark.parametrize("seed", [1, 3, 4, 5, 6, 8])
def test_random_programs_are_accepted(seed):
    blob, ins = pg.random_program(seed, hashes=False)
    ores, pub = _case(blob, ins)
    assert sum(1 for w in ores.rows["instruction"] if 0x30 <= (int(w) & 0x7F) <= 0x3B) > 10
    assert so.verify(so.prove(ores.rows, pub), pub) == 0


@pytest.mark.parametrize("name,check", [("self_modifying", 55), ("sha256_hello", 10), ("mem_sw_lw", 55), ("timestamps", 55), ("rc_doubling", 55), ("q9_access_at_own_pc", 55)])
def test_runs_outside_the_air_have_no_proof(name, check):
    """A hash syscall writes memory the AIR does not state: rejected (10).  An access to the CODE SEGMENT (format v11, check 55): instruction fetch is tied to the
    program's words, so a store into the code would change what the VM executes (strict protection is off, vm.rs:175) but not what the AIR lets through — a mode-3 proof is of a
    run whose loads and stores stay off the cells that overlap the code, and the verifier checks the touched cells the proof carries.  That refuses the reference's own memory
    tests as written (they store at 0x1000, the first code word: vm.rs:996-1200, :698-752) and the Q9 program (an access at its own pc); off the code they are proven above."""
    blob, ins, cfg = getattr(pg, name)()
    ores, pub = _case(blob, ins)
    assert so.verify(so.prove(ores.rows, pub), pub) == check


def _two_stores_one_load():
    code = [A(3, 0, 0x4000), A(1, 0, 0x1111), A(2, 0, 0x2222), spec.sw(3, 1, 0), spec.sw(3, 2, 0), spec.lw(4, 3, 0), A(7, 0, 1), pg.EB]
    blob = pg._p(code)
    ores, pub = _case(blob)
    return ores, pub, so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)


def test_a_stale_read_breaks_only_the_memory_check():
    """SW 0x1111, SW 0x2222, LW: a prover lets the load return the FIRST store's value (old bytes, old time, pieces, y and the register afterwards all consistent).  Every
    local constraint holds — what fails is the closing of the running sum: the multiset {initial} + {written} != {read} + {final}."""
    ores, pub, M, cells = _two_stores_one_load()
    assert [int(x) for x in cells[0]] == [0x4000, 0, 6, 0x2222, 0, 0, 0]
    assert np.array_equal(so.prove_matrix_mem(M, pub, cells), so.prove(ores.rows, pub))
    F, i = M.copy(), 5
    F[C_OB, i] = F[C_OB + 1, i] = 0x11
    F[C_TOLD, i] = M[C_TOLD, 4]
    F[C_PIECE, i] = F[C_PIECE + 1, i] = 0x11
    F[C_Y, i] = 0x1111
    F[C_LIMB + 3 * 4, i + 1:] = 0x1111
    dt = int(F[C_CYCLE, i]) - int(F[C_TOLD, i])
    F[C_RC2, i], F[C_RC2 + 1, i], F[C_RC2 + 2, i] = dt & 1023, (dt >> 10) & 1023, dt >> 20
    n, bad = so.failing_constraints(F, pub, cells)
    assert n == 4 and [c for c, _ in bad] == [I_SUM, I_SUM + 1, I_SUM + 2, I_SUM + 3]
    assert so.verify(so.prove_matrix_mem(F, pub, cells), None) == 10
    c2 = cells.copy(); c2[0, 3] = 0x1111                                           # .. and no choice of final cell repairs it
    assert so.verify(so.prove_matrix_mem(F, pub, c2), None) == 10
    c3 = cells.copy(); c3[0, 2] = 5
    assert so.verify(so.prove_matrix_mem(F, pub, c3), None) == 10


def test_forged_cells_and_values_are_rejected():
    ores, pub, M, cells = _two_stores_one_load()
    c = cells.copy(); c[0, 3] ^= 1
    assert so.verify(so.prove_matrix_mem(M, pub, c), None) == 10                   # forged final bytes
    assert so.verify(so.prove_matrix_mem(M, pub, cells[:0]), None) == 10           # a touched cell left out
    assert so.verify(so.prove_matrix_mem(M, pub, np.concatenate([cells, cells])), None) == 54      # a cell listed twice (two initial tuples)
    c = cells.copy(); c[0, 0] += 4
    assert so.verify(so.prove_matrix_mem(M, pub, c), None) == 54                   # not a cell address
    G = M.copy(); G[C_Y, 5] = 0x2223; G[C_LIMB + 12, 6:] = 0x2223                  # the load writes something else than the cell holds
    assert so.failing_constraints(G, pub, cells)[0] == 1
    assert so.verify(so.prove_matrix_mem(G, pub, cells), None) == 10
    proof = so.prove(ores.rows, pub)
    at = int(np.nonzero(proof == 0x2222)[0][0])                                    # the final bytes in the proof's memory section: bound by the transcript
    t = proof.copy(); t[at] = 0x2223
    assert so.verify(t, None) != 0


def test_a_write_in_the_future_cannot_be_read():
    """The time read must be SMALLER than the time written: a load that claims to read what a later store writes violates the range check on cycle - told."""
    code = [A(3, 0, 0x4000), A(1, 0, 0x77), spec.lw(4, 3, 0), spec.sw(3, 1, 0), A(7, 0, 1), pg.EB]
    ores, pub = _case(pg._p(code))
    M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    F = M.copy()
    # the load (row 2) reads the tuple the store (row 3, time 4) writes; the store reads the initial tuple
    F[C_OB, 2] = 0x77; F[C_TOLD, 2] = 4; F[C_PIECE, 2] = 0x77; F[C_Y, 2] = 0x77; F[C_LIMB + 12, 3:] = 0x77
    F[C_OB, 3] = 0; F[C_TOLD, 3] = 0
    dt = (int(F[C_CYCLE, 2]) - 4) % so.P
    F[C_RC2, 2], F[C_RC2 + 1, 2], F[C_RC2 + 2, 2] = dt & 1023, (dt >> 10) & 1023, dt >> 20      # not three 10-bit chunks any more
    c = cells.copy(); c[0, 2] = 3
    assert so.verify(so.prove_matrix_mem(F, pub, c), None) == 10


def test_bitwise_opcodes_are_constrained():
    """AND OR XOR ANDI ORI XORI on 40-bit values with bits above 40 in the registers (masked: Value40::from_u64) and negative immediates (sign-extended, then masked): honest
    rows satisfy every constraint; a wrong result nibble, a result of the wrong operation, an operand nibble that is not the register's, a tuple smuggled onto another row —
    each violates a constraint or leaves its table (the multiplicity is then not counted and the running sum does not close)."""
    O, E = spec.Opcode, spec.encode
    code = pg.li40(1, 0xF0F0A5C3E1) + pg.li40(2, 0x0FF0FF00FF) + [A(5, 0, 0x4000), A(6, 0, -1), E(O.SD, rs1=5, rs2=6, imm=0), E(O.LB, 7, 5, imm=0)]      # r7 = 0xFFFF..FF (64 bits)
    code += [E(O.AND, 8, 1, 2), E(O.OR, 9, 1, 2), E(O.XOR, 10, 1, 2), E(O.ANDI, 11, 1, imm=-256), E(O.ORI, 12, 1, imm=0x5555), E(O.XORI, 13, 1, imm=-1),
             E(O.AND, 14, 7, 1), E(O.XOR, 15, 7, 7), E(O.OR, 0, 1, 2), pg.EB]
    ores, pub = _case(pg._p(code))
    M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    ops = ores.rows["instruction"] & 0x7F
    rows = {int(o): int(np.nonzero(ops == o)[0][-1]) for o in (0x10, 0x11, 0x12, 0x13, 0x14, 0x15)}
    a, b = 0xF0F0A5C3E1, 0x0FF0FF00FF
    nxt = ores.rows["registers"]
    assert int(nxt[rows[0x12] + 1][10]) == a ^ b and int(nxt[rows[0x13] + 1][11]) == a & 0xFFFFFFFF00 and int(nxt[rows[0x15] + 1][13]) == a ^ 0xFFFFFFFFFF
    assert int(nxt[rows[0x10] + 1][14]) == a                                       # AND with 0xFFFF_FFFF_FFFF_FFFF masked to 40 bits
    assert so.failing_constraints(M, pub, cells)[0] == 0 and so.verify(so.prove(ores.rows, pub), pub) == 0
    C_KLG, C_OA, C_OO, C_LB, C_LR, C_PIECE = 220, 221, 222, 224, 234, 206

    def bad(edit):
        F = M.copy(); edit(F)
        return so.failing_constraints(F, pub, cells)[0] > 0 and so.verify(so.prove_matrix_mem(F, pub, cells), None) == 10
    i = rows[0x12]
    assert bad(lambda F: F.__setitem__((C_LR + 3, i), int(F[C_LR + 3, i]) ^ 1))                                             # a wrong result nibble (y no longer matches)
    def wrong_nibble_consistent(F):                                                                                          # .. with y and the register following it
        F[C_LR, i] = int(F[C_LR, i]) ^ 1; F[C_Y, i] = int(F[C_Y, i]) ^ 1; F[C_LIMB + 30, i + 1:] = F[C_Y, i]
    assert bad(wrong_nibble_consistent)                                                                                     # only the table lookup catches this one
    def or_as_xor(F):                                                                                                       # an XOR row computing OR: every nibble tuple is in the OR table, not in XOR's
        v = a | b
        for k in range(10): F[C_LR + k, i] = (v >> (4 * k)) & 15
        F[C_Y, i], F[C_Y + 1, i] = v & 0xFFFFF, v >> 20; F[C_LIMB + 30, i + 1:] = v & 0xFFFFF; F[C_LIMB + 31, i + 1:] = v >> 20
    assert bad(or_as_xor)
    assert bad(lambda F: (F.__setitem__((C_OA, i), 1)))                                                                     # two operations at once
    assert bad(lambda F: F.__setitem__((C_PIECE + 2, i), (int(F[C_PIECE + 2, i]) + 1) % 16))                                # an operand nibble that is not rs1's
    j = rows[0x13]
    assert bad(lambda F: F.__setitem__((C_LB + 1, j), (int(F[C_LB + 1, j]) + 1) % 16))                                      # .. that is not the immediate's
    k0 = int(np.nonzero(ops == 0x08)[0][0])
    assert bad(lambda F: F.__setitem__((C_LB, k0), 3))                                                                      # a tuple element on a row that is no bitwise row


def test_shifts_are_constrained():
    """SLL SRL SRA SLLI SRLI SRAI on 40-bit values (value.rs:658-697): every amount class — 0, inside a chunk, on chunk borders, 39, 40 (everything shifted out), beyond 40 and
    beyond 49, a register amount with bits above 63 set (masked), an 8-bit shamt of 255 — on a negative and a positive value; honest rows satisfy every constraint; a result off by
    a bit, a shift by another amount than the register says, a left shift computed as a right shift, a forged sign fill: each is rejected."""
    O, E = spec.Opcode, spec.encode
    code = pg.li40(1, 0xF0F0A5C3E1) + pg.li40(2, 0x0312345678)
    amounts = [0, 1, 9, 10, 11, 19, 20, 29, 30, 39, 40, 41, 49, 50, 63]
    for n, sh in enumerate(amounts):
        code += [A(3, 0, sh + (64 if n % 2 else 0)), E(O.SLL, 4, 1, 3), E(O.SRL, 5, 1, 3), E(O.SRA, 6, 1, 3), E(O.SRA, 7, 2, 3)]
    for sh in amounts + [64, 200, 255]:
        code += [E(O.SLLI, 8, 1, imm=sh), E(O.SRLI, 9, 1, imm=sh), E(O.SRAI, 10, 1, imm=sh), E(O.SRAI, 11, 2, imm=sh), E(O.SLLI, 0, 2, imm=sh)]
    ores, pub = _case(pg._p(code + [pg.EB]))
    M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    regs, ops = ores.rows["registers"], ores.rows["instruction"] & 0x7F
    a, m40 = 0xF0F0A5C3E1, (1 << 40) - 1
    for i in np.nonzero((ops >= 0x18) & (ops <= 0x1D))[0]:                                  # the VM's own results are what value.rs says (the oracle is the restated VM)
        w = int(ores.rows["instruction"][i]); op = w & 0x7F; rd = (w >> 7) & 15; rs1 = (w >> 11) & 15
        sh = ((w >> 15) & 0xFF) if op >= 0x1B else int(regs[i][(w >> 15) & 15]) & 63
        v = int(regs[i][rs1]) & m40
        want = {0: 0 if sh >= 40 else (v << sh) & m40, 1: 0 if sh >= 40 else v >> sh,
                2: ((m40 if v >> 39 else 0) if sh >= 40 else ((v >> sh) | ((m40 ^ ((1 << (40 - sh)) - 1)) if v >> 39 else 0)))}[(op - 0x18) % 3]
        if rd:
            assert int(regs[i + 1][rd]) == want, (hex(op), sh, hex(v))
    assert so.failing_constraints(M, pub, cells)[0] == 0 and so.verify(so.prove(ores.rows, pub), pub) == 0
    C_KSH, C_UL, C_UR, C_V, C_SGN, C_PR, C_ON, C_SH = 244, 245, 250, 255, 268, 269, 273, 275

    def bad(edit):
        F = M.copy(); edit(F)
        return so.failing_constraints(F, pub, cells)[0] > 0 and so.verify(so.prove_matrix_mem(F, pub, cells), None) == 10
    nr = len(ops)
    i = int(np.nonzero((ops == 0x18) & (M[C_SH][:nr] == 11))[0][0])                              # SLL by 11 (u = 1, v = 1)
    assert M[C_UL + 1, i] == 1 and M[C_V + 1, i] == 1

    def result_off_by_a_bit(F):
        F[C_Y, i] = int(F[C_Y, i]) ^ 2; F[C_LIMB + 12, i + 1:i + 2] = F[C_Y, i]
    assert bad(result_off_by_a_bit)

    def other_amount(F):                                                                   # the row shifts by 12 while the register says 11
        F[C_V + 1, i] = 0; F[C_V + 2, i] = 1
        a_ = int(regs[i][1]) & m40
        for k in range(4):
            pr = ((a_ >> (10 * k)) & 1023) << 2
            F[C_PR + k, i] = pr; F[135 + k, i] = pr & 1023; F[C_PIECE + k, i] = pr >> 10
        r = (a_ << 12) & m40
        F[C_Y, i], F[C_Y + 1, i] = r & 0xFFFFF, r >> 20; F[C_LIMB + 12, i + 1:i + 2] = r & 0xFFFFF; F[C_LIMB + 13, i + 1:i + 2] = r >> 20
    assert bad(other_amount)
    assert bad(lambda F: (F.__setitem__((C_UL + 1, i), 0), F.__setitem__((C_UR + 1, i), 1)))   # a left shift run as a right shift (the opcode says left)
    j = int(np.nonzero((ops == 0x1A) & (M[C_SH][:nr] == 9) & (M[C_SGN][:nr] == 1))[0][0])             # SRA by 9 of a negative value
    assert bad(lambda F: F.__setitem__((C_SGN, j), 0))                                       # the sign fill dropped
    assert bad(lambda F: F.__setitem__((C_ON, j), int(F[C_ON, j]) ^ 1))                      # .. or of another width
    k0 = int(np.nonzero(ops == 0x08)[0][0])
    assert bad(lambda F: F.__setitem__((C_V + 3, k0), 1))                                    # a bit shift on a row that is no shift


def test_mul_is_constrained():
    """MUL = Value40::wrapping_mul of the masked operands (execute.rs:79-99), a schoolbook product in 10-bit chunks: honest rows over a grid of extreme operands satisfy every
    constraint and the VM's results are the products mod 2^40; a result off by one (consistently carried into the register), a forged carry, a chunk of the wrong operand, a
    carry bit that is no bit, a MUL run on a row that is none: each is rejected."""
    blob, ins, _ = pg.mul_grid()
    ores, pub = _case(blob)
    M, cells = so.main_trace(ores.rows, pub), so.mem_cells(ores.rows, pub)
    regs, words = ores.rows["registers"], ores.rows["instruction"]
    ops = words & 0x7F
    m40 = (1 << 40) - 1
    rows = np.nonzero(ops == 0x02)[0]
    assert len(rows) > 400
    for i in rows:
        w = int(words[i]); rd = (w >> 7) & 15
        want = ((int(regs[i][(w >> 11) & 15]) & m40) * (int(regs[i][(w >> 15) & 15]) & m40)) & m40
        if rd:
            assert int(regs[i + 1][rd]) == want
    assert so.failing_constraints(M, pub, cells)[0] == 0 and so.verify(so.prove(ores.rows, pub), pub) == 0
    C_KMU, C_MA, C_ME, C_RC = 276, 277, 281, 135
    nr = len(ops)
    assert np.array_equal(M[C_KMU][:nr] != 0, (ops == 0x02) & (np.arange(nr) < nr - 1))
    assert M[C_ME][rows].max() == 1 and (M[C_ME + 1][rows] + 2 * M[C_ME + 2][rows]).max() == 2 and M[C_PIECE + 8][rows].max() == 3      # the largest carries occur

    def bad(edit):
        F = M.copy(); edit(F)
        return so.failing_constraints(F, pub, cells)[0] > 0 and so.verify(so.prove_matrix_mem(F, pub, cells), None) == 10
    i = int([r for r in rows if ((int(words[r]) >> 7) & 15) == 3 and int(regs[r][1]) == 0xF0F0A5C3E1 and int(regs[r][2]) == 0x0312345678][0])

    def result_off_by_one(F):                                                               # y, its chunk and the register that follows agree with each other — not with the product
        F[C_Y, i] = int(F[C_Y, i]) ^ 1; F[C_RC, i] = int(F[C_RC, i]) ^ 1; F[C_LIMB + 9, i + 1:i + 2] = F[C_Y, i]
    assert bad(result_off_by_one)
    assert bad(lambda F: F.__setitem__((C_PIECE + 5, i), (int(F[C_PIECE + 5, i]) + 1) % 1024))                # a forged carry
    assert bad(lambda F: F.__setitem__((C_MA + 2, i), (int(F[C_MA + 2, i]) + 1) % 1024))                      # ma_2 is not kmu a_2
    assert bad(lambda F: (F.__setitem__((C_PIECE + 1, i), (int(F[C_PIECE + 1, i]) + 1) % 1024)))             # a chunk that is not rs2's
    assert bad(lambda F: F.__setitem__((C_ME, i), 2))                                                        # a carry "bit" of 2
    k0 = int(np.nonzero(ops == 0x08)[0][0])
    assert bad(lambda F: F.__setitem__((C_KMU, k0), 1))                                                      # an ADDI run as a MUL
    assert bad(lambda F: F.__setitem__((C_MA, k0), 5))                                                       # a's chunks off the MUL rows


# ---- the product's verifier (zkir_verify, verify.cpp + air.h) on the oracle's mode-3 proofs: same verdict and same failing check ------------------------------------
def _pub_c(p):
    from zkir_amd import runtime as rt
    out = rt.PublicInputsC(p.n_real, p.entry, p.deferred, 0)
    out.program_digest[:] = list(p.prog); out.io_digest[:] = list(p.io)
    return out


@pytest.mark.parametrize("name", ["timestamps", "loads_stores", "echo5", "random3", "mem_sw_lw_on_code"])
def test_product_verifier_agrees_with_the_oracle(name):
    from zkir_amd import runtime as rt
    if name == "mem_sw_lw_on_code":                     # the reference's test as written (a store at 0x1000): both verifiers refuse the touched code cell (55)
        blob, ins, cfg = pg.mem_sw_lw()
        ores, pub = _case(blob, ins)
        pr = so.prove(ores.rows, pub)
        assert so.verify(pr, pub) == 55 and rt.verify(pr) == 55 and rt.verify(pr, _pub_c(pub)) == 55
        return
    if name.startswith("random"):
        blob, ins = pg.random_program(int(name[6:]), hashes=False); cfg = {}
    else:
        blob, ins, cfg = pg.off_code(name)
    ores, pub = _case(blob, ins)
    pr = so.prove(ores.rows, pub)
    assert so.verify(pr, pub) == 0
    assert rt.verify(pr) == 0 and rt.verify(pr, _pub_c(pub)) == 0
    assert rt.verify_segment(pr)[0] == 2
    rng = np.random.default_rng(len(pr))
    hw = 157 + 4
    blob_words = (len(blob) + 1) // 2
    for pos in list(range(2, 21)) + [hw - 1, hw + 3] + [int(x) for x in rng.integers(hw + 1 + blob_words, len(pr), 80)] + [len(pr) - 1]:
        t = pr.copy()
        t[pos] = (int(t[pos]) + 1 + int(rng.integers(0, 50))) % so.P
        if t[pos] != pr[pos]:
            want = so.verify(t)
            assert want != 0 and rt.verify(t) == want, (pos, want, rt.verify(t))
    assert rt.verify(pr[:-1]) == so.verify(pr[:-1]) != 0


def test_product_verifier_rejects_what_the_oracle_rejects():
    """Cheating provers' mode-3 proofs (made by the oracle prover from forged matrices / cells): the product's verifier gives the oracle's verdict."""
    from zkir_amd import runtime as rt
    ores, pub, M, cells = _two_stores_one_load()
    F, i = M.copy(), 5
    F[C_OB, i] = F[C_OB + 1, i] = 0x11; F[C_TOLD, i] = M[C_TOLD, 4]; F[C_PIECE, i] = F[C_PIECE + 1, i] = 0x11; F[C_Y, i] = 0x1111; F[C_LIMB + 12, i + 1:] = 0x1111
    dt = int(F[C_CYCLE, i]) - int(F[C_TOLD, i])
    F[C_RC2, i], F[C_RC2 + 1, i], F[C_RC2 + 2, i] = dt & 1023, (dt >> 10) & 1023, dt >> 20
    for proof, want in [(so.prove_matrix_mem(F, pub, cells), 10), (so.prove_matrix_mem(M, pub, cells[:0]), 10), (so.prove_matrix_mem(M, pub, np.concatenate([cells, cells])), 54)]:
        assert so.verify(proof, None) == want and rt.verify(proof) == want
    c = cells.copy(); c[0, 3] ^= 1
    proof = so.prove_matrix_mem(M, pub, c)
    assert so.verify(proof, None) == rt.verify(proof) == 10


def test_a_run_cut_by_its_cycle_limit_on_a_write_is_accepted():
    """vm.rs:211-214, :302-347: a run that stops at its cycle limit has EXECUTED its last row (cycles == rows).  If that row is a WRITE ecall its output exists, while the AIR's
    counters say what happened before a row: both verifiers read the last output off the public last state (R10 = 2, R11 = the value) — modes 2 and 3; a forged last output is 51."""
    from zkir_amd import runtime as rt
    code = [A(1, 0, 7), A(11, 1, 0), A(10, 0, 2), spec.ecall(), A(1, 1, 1), spec.jal(0, -16)]           # WRITE 7, 8, 9, .. forever
    blob = pg._p(code)
    for n, n_out in ((4, 1), (9, 2), (10, 2), (14, 3)):                                                # rows 3, 8, 13 are the WRITEs: n = 4, 9, 14 end ON one
        ores = oracle.run(blob, [], max_cycles=n, enable_execution_trace=True)
        assert len(ores.rows) == n and len(ores.outputs) == n_out and ores.halt_kind == 2
        for kw in (dict(io_mode=True), dict(mem_mode=True)):
            pub = so.public_inputs(n, blob, [], list(ores.outputs), (2, 0), **kw)
            pr = so.prove(ores.rows, pub)
            assert so.verify(pr, pub) == 0 and rt.verify(pr) == 0, (n, kw)
            if n in (4, 9, 14):
                fake = so.public_inputs(n, blob, [], list(ores.outputs[:-1]) + [int(ores.outputs[-1]) + 1], (2, 0), **kw)
                fp = so.prove(ores.rows, fake)
                assert so.verify(fp, fake) == 51 and rt.verify(fp) == 51


# ---- format v11 (round 5): EBREAK is a class of its own in modes 2 / 3 ------------------------------------------------------------------------------
def test_an_ebreak_cannot_be_stepped_over():
    """The VM halts on EBREAK (execute.rs:667).  A cheating prover's trace steps OVER one — an assert / abort path — as if it were a sequential instruction, and goes on to a
    different exit.  In modes 0 / 1 (format v10) the EBREAK word is class "other" and the trace is accepted: those modes state the control flow, not every opcode.  In modes 2 / 3
    the word's class id (21) is carried by no selector, so constraint 4 cannot hold on an executed EBREAK row: only the halt row can sit on it — both verifiers say 10."""
    from zkir_amd import runtime as rt
    MULH = 0x03                                                          # class "other", writes nothing with rd = 0: what the cheating trace is made from
    tail = [A(11, 0, 7), A(10, 0, 2), spec.ecall(), A(10, 0, 0), A(11, 0, 0), spec.ecall()]       # WRITE 7; EXIT 0
    prog = pg._p([A(1, 0, 5), pg.EB] + tail)                             # the program: aborts at its second word
    twin = pg._p([A(1, 0, 5), MULH] + tail)                              # .. and the program whose run the cheater copies
    honest = oracle.run(prog, [], enable_execution_trace=True)
    assert len(honest.rows) == 2 and honest.halt_kind == 0 and list(honest.outputs) == []
    run = oracle.run(twin, [], enable_execution_trace=True)
    assert len(run.rows) == 8 and list(run.outputs) == [7] and (run.halt_kind, run.halt_code) == (1, 0)
    rows = run.rows.copy()
    rows["instruction"][1] = pg.EB                                       # the forged trace: the run of `twin`, labelled with `prog`'s words
    for kw, verdict in ((dict(), 0), (dict(io_mode=True), 10), (dict(mem_mode=True), 10)):
        pub = so.public_inputs(len(rows), prog, [], [7], (1, 0), **kw)
        pr = so.prove(rows, pub)
        assert so.verify(pr, pub) == verdict and rt.verify(pr) == verdict, kw
    for kw in (dict(io_mode=True), dict(mem_mode=True)):                 # the honest run (halting ON the EBREAK) is proven as before
        pub = so.public_inputs(2, prog, [], [], (0, 0), **kw)
        pr = so.prove(honest.rows, pub)
        assert so.verify(pr, pub) == 0 and rt.verify(pr) == 0 and pr[1] == 11


def test_a_mode2_segment_binds_its_tapes():
    """(v11) The I/O section of a mode-2 proof enters the transcript before the lookup challenges — a SEGMENT's too, whose io digest only the chain checks: a
    segment proof whose tapes were changed after the fact is rejected on its own (before: zkir_verify_segment returned 0 for it)."""
    from zkir_amd import runtime as rt
    blob, ins, cfg = pg.echo5()
    ores = oracle.run(blob, list(ins), enable_execution_trace=True)
    n, cut = len(ores.rows), len(ores.rows) // 2
    w0 = sum(1 for r in ores.rows[:cut] if (int(r["instruction"]) & 0x7F) == 0x50 and int(r["registers"][10]) == 2)
    r0 = sum(1 for r in ores.rows[:cut] if (int(r["instruction"]) & 0x7F) == 0x50 and int(r["registers"][10]) == 1)
    pubs = [so.public_inputs(cut + 1, blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), io_mode=True),
            so.public_inputs(n - cut, blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), io_mode=True, writes_before=w0, reads_before=r0)]
    for p in pubs:
        p.io[:] = list(so.public_inputs(n, blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), io_mode=True).io)
    segs = [so.prove(ores.rows[:cut + 1], pubs[0]), so.prove(ores.rows[cut:], pubs[1])]
    run_pub = so.public_inputs(n, blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), io_mode=True)
    assert so.verify_chain(segs, run_pub) == 0 and rt.verify_chain(segs) == 0
    for s in segs:
        assert so.verify_segment(s)[0] == 0 and rt.verify_segment(s)[0] == 0
        lay_hw = 157 + 4
        at = lay_hw + 1 + (int(s[lay_hw]) + 1) // 2                      # the I/O section: [n_in] [inputs ..] [n_out] [outputs ..] [halt kind] [halt code]
        t = s.copy()
        t[at + 1] = (int(t[at + 1]) + 1) & 0xFFFF                        # the first input's low piece
        assert so.verify_segment(t)[0] != 0 and rt.verify_segment(t)[0] == so.verify_segment(t)[0]


def test_proof_layout_walks_every_mode():
    """zkir_amd.stark.proof_layout (it assumed a mode-0 header): on proofs of modes 0, 2 and 3 the trace root it finds is the commitment of the run's trace, the
    program it finds is the program, and the sections it skips are where the oracle put them."""
    from zkir_amd import stark
    blob, ins, cfg = pg.echo5()
    ores = oracle.run(blob, list(ins), enable_execution_trace=True)
    for kw, mode in ((dict(), 0), (dict(io_mode=True), 2), (dict(mem_mode=True), 3)):
        pub = so.public_inputs(len(ores.rows), blob, list(ins), list(ores.outputs), (ores.halt_kind, ores.halt_code), **kw)
        pr = so.prove(ores.rows, pub)
        lay = stark.proof_layout(pr)
        assert lay["mode"] == mode and lay["blob"] == blob and (lay["num_queries"], lay["pow_bits"]) == (50, 12)
        assert stark.trace_root(pr) == [int(x) for x in so.commit_trace(ores.rows, 1, pub=pub)]
        if mode >= 2:
            at = lay["io_section"]
            assert int(pr[at]) == len(ins) and int(pr[at + 1 + 4 * len(ins)]) == len(ores.outputs)
        if mode == 3:
            assert int(pr[lay["mem_section"]]) == len(so.mem_cells(ores.rows, pub))
