"""Test programs: the reference's own test programs re-typed with the v3.4 encoder, targeted quirk probes
 and a seeded random-program generator.  Each entry returns (blob, inputs, cfg_kwargs)."""
from __future__ import annotations

import numpy as np

from zkir_amd import spec
from zkir_amd.spec import Opcode as O, encode as E, Program

A = lambda rd, rs1, imm: E(O.ADDI, rd, rs1, imm=imm)  # noqa: E731
EB = spec.ebreak()
EC = spec.ecall()


def _p(code, data=b"", config=None):
    return Program.from_code(code, data, config).to_bytes()


def _exit0():
    return [A(10, 0, 0), A(11, 0, 0), EC]


def li40(rd, value, tmp=None):
    """Load an arbitrary 40-bit constant with 16-bit pieces (17-bit signed immediates, Q11)."""
    v = value & ((1 << 40) - 1)
    out = [A(rd, 0, (v >> 32) & 0xFF)]
    for sh in (16, 0):
        out += [E(O.SLLI, rd, rd, imm=16), E(O.ORI, rd, rd, imm=(v >> sh) & 0xFFFF)]
    return out


# ---- the reference's own programs ---------------------------------------------------------------
def exit_42():                      # vm.rs:464-486
    return _p([A(10, 0, 0), A(11, 0, 42), EC]), [], {}


def io_echo():                      # vm.rs:489-533
    return _p([A(10, 0, 1), EC, A(11, 10, 0), A(10, 0, 2), EC, A(11, 0, 0), A(10, 0, 0), EC]), [123], {}


def echo5():                        # stress_tests.rs:397-430 (five values, exhausted tape reads 0)
    code = []
    for _ in range(6):
        code += [A(10, 0, 1), EC, A(11, 10, 0), A(10, 0, 2), EC]
    return _p(code + _exit0()), [1, 2, 3, 4, 5], {}


def jal_self_cycle_limit():         # vm.rs:536-552
    return _p([spec.jal(0, 0)]), [], {"max_cycles": 100}


def basic_add_ebreak():             # vm.rs:434-461
    return _p([A(1, 0, 10), A(2, 0, 20), spec.add(3, 1, 2), EB]), [], {}


# The reference's own memory tests store at 0x1000 — the FIRST CODE WORD (strict protection is off, vm.rs:175; the word has been executed by then).  `base` moves the data
# off the code segment: what a mode-3 proof needs (format v11: no touched cell may overlap the code, check 55) — OFF_CODE below.
OFF_CODE = 0x2000


def mem_sw_lw(base=0x1000):         # vm.rs:996-1070
    return _p([A(1, 0, 0x42), A(3, 0, base), spec.sw(3, 1, 0), spec.lw(4, 3, 0), EB]), [], {}


def timestamps(base=0x1000):        # vm.rs:1073-1200
    return _p([A(1, 0, 0x100), A(2, 0, base), spec.sw(2, 1, 0), A(3, 0, 0x200), spec.sw(2, 3, 4), spec.lw(4, 2, 0),
               spec.lw(5, 2, 4), EB]), [], {}


def beq_skip():                     # vm.rs:594-632
    return _p([A(1, 0, 10), A(2, 0, 10), spec.beq(1, 2, 8), A(3, 0, 99), EB]), [], {}


def sum_1_to_5():                   # cross_module.rs:406-437
    return _p([A(1, 0, 0), A(2, 0, 1), A(3, 0, 6), spec.add(1, 1, 2), A(2, 2, 1), spec.bne(2, 3, -8), A(11, 1, 0), A(10, 0, 2), EC]
              + _exit0()), [], {}


def fib30():
    return spec.fib_program(30).to_bytes(), [], {}


def rc_doubling(base=0x1000):       # vm.rs:698-752: 30 doublings then SW -> deferred range checks flushed
    code = [A(1, 0, (1 << 15) - 1)] + [spec.add(1, 1, 1)] * 30 + [A(2, 0, base), spec.sw(2, 1, 0), EB]
    return _p(code), [], {"enable_range_checking": True}


def rc_small_consts():              # vm.rs:755-806: no witnesses
    return _p([A(1, 0, 100), A(2, 0, 200), spec.add(3, 1, 2), A(4, 0, 0x2000), spec.sw(4, 3, 0), EB]), [], {"enable_range_checking": True}


def rc_many_pending():              # >= 16 pending checks force a checkpoint without an observation opcode (range_check.rs:122-135)
    code = [A(1, 0, 3)] + li40(2, 0xFFFFFFFFFF)
    code += [spec.mul(3, 2, 2)] * 3 + [spec.add(4, 3, 3)] * 40 + [spec.mul(5, 4, 2)] * 5 + [spec.bne(0, 0, 8), EB]
    return _p(code), [], {"enable_range_checking": True}


def rc_config_30bit():              # header limb_bits=30 -> chunk_bits=15, data_bits=60 (vm.rs:185, range_check.rs:29)
    code = li40(2, 0xABCDE12345) + [spec.mul(3, 2, 2), spec.mul(4, 3, 3), spec.mul(5, 4, 4), A(6, 0, 0x2000), spec.sw(6, 5, 0), EB]
    return _p(code, config=spec.Config(30, 2, 2)), [], {"enable_range_checking": True}


# ---- ALU / bound-algebra coverage ---------------------------------------------------------------
def alu_all():
    code = li40(1, 0xFEDCBA9876) + li40(2, 0x0123456789) + [A(3, 0, -5), A(4, 0, 37), A(5, 0, 1)]
    for op in (O.ADD, O.SUB, O.MUL, O.MULH, O.DIVU, O.REMU, O.DIV, O.REM, O.AND, O.OR, O.XOR, O.SLL, O.SRL, O.SRA, O.SLTU, O.SGEU,
               O.SLT, O.SGE, O.SEQ, O.SNE, O.CMOV, O.CMOVZ, O.CMOVNZ):
        for (a, b) in ((1, 2), (2, 1), (3, 4), (1, 3), (3, 3), (4, 5), (0, 1), (1, 0) if op not in (O.DIVU, O.REMU, O.DIV, O.REM) else (1, 5)):
            code.append(E(op, 6 + (len(code) % 4), a, b))
    for op in (O.ADDI, O.ANDI, O.ORI, O.XORI):
        for imm in (0, 1, -1, 65535, -65536, 0x7FFF, -12345):
            code.append(E(op, 6 + (len(code) % 4), 1 + (len(code) % 3), imm=imm))
    for op in (O.SLLI, O.SRLI, O.SRAI):
        for sh in (0, 1, 19, 20, 39, 40, 41, 63, 64, 255):
            code.append(E(op, 6 + (len(code) % 4), 1 + (len(code) % 3), imm=sh))
    # shifts by register use rs2 & 0x3F (Q5)
    code += [A(9, 0, 64 + 3), E(O.SLL, 6, 1, 9), E(O.SRL, 7, 1, 9), E(O.SRA, 8, 1, 9), A(9, 0, 45), E(O.SLL, 6, 1, 9), E(O.SRA, 8, 1, 9)]
    return _p(code + [EB]), [], {}


def mul_grid():
    """MUL (execute.rs:79-99) on a grid of 40-bit operands: zero, one, all ones, the sign bit, every chunk at its extremes (the largest carries), a register with bits above 40
    set by LB's sign extension (masked: Value40::from_u64), rs1 = rs2 = rd."""
    vals = [0, 1, 2, 0xFFFFFFFFFF, 0x8000000000, 0xF0F0A5C3E1, 0x0312345678, 1023, 1024, 0xFFFFF, 0x100000, 0x3FF003FF, 0xFFC00FFC00, 0x7FFFFFFFFF]
    code = [A(5, 0, 0x4000), A(6, 0, 0x80), E(O.SB, rs1=5, rs2=6, imm=0), E(O.LB, 7, 5, imm=0)]           # R7 = 0xFFFF...FF80 (Q1)
    for a in vals:
        code += li40(1, a)
        for b in vals:
            code += li40(2, b) + [E(O.MUL, 3, 1, 2), E(O.MUL, 4, 2, 1)]
        code += [E(O.MUL, 8, 1, 7), E(O.MUL, 9, 7, 1), E(O.MUL, 1, 1, 1)]
    code += [E(O.MUL, 7, 7, 7), E(O.MUL, 0, 1, 2)]
    return _p(code + [EB]), [], {}


def wide_grid(vals=None):
    """MULH DIVU REMU DIV REM (execute.rs:101-183) on a grid of operands BELOW 2^40 (AIR mode 4's domain): zero and one, all ones, the sign bit of a 40-bit value (positive as an
    i64: quirk Q2), every chunk at its extremes (the largest carries of the 80-bit product), dividend < divisor, equal operands, powers of two, rs1 = rs2 = rd, rd = r0."""
    vals = vals or [0, 1, 2, 0xFFFFFFFFFF, 0x8000000000, 0xF0F0A5C3E1, 0x0312345678, 1023, 1024, 0xFFFFF, 0x100000, 0x3FF003FF, 0xFFC00FFC00, 0x7FFFFFFFFF]
    code = []
    for a in vals:
        code += li40(1, a)
        for b in vals:
            code += li40(2, b) + [E(O.MULH, 3, 1, 2), E(O.MULH, 4, 2, 1)]
            if b:
                code += [E(O.DIVU, 5, 1, 2), E(O.REMU, 6, 1, 2), E(O.DIV, 7, 1, 2), E(O.REM, 8, 1, 2)]
        code += [E(O.MULH, 9, 1, 1)] + ([E(O.DIVU, 9, 1, 1), E(O.REM, 0, 1, 1)] if a else [])
    code += li40(1, 0xFEDCBA9876) + [E(O.REMU, 1, 1, 1)] + li40(1, 0xFEDCBA9876) + [E(O.DIV, 1, 1, 1), E(O.MULH, 1, 1, 1)]
    return _p(code + [EB]), [], {}


def loads_stores():
    code = [A(5, 0, 0x4000)] + li40(1, 0x80F1E2D3C4) + [A(2, 0, -1)]
    code += [E(O.SD, rs1=5, rs2=1, imm=0), E(O.SW, rs1=5, rs2=1, imm=8), E(O.SH, rs1=5, rs2=1, imm=12), E(O.SB, rs1=5, rs2=1, imm=14),
             E(O.SD, rs1=5, rs2=2, imm=16), E(O.SB, rs1=5, rs2=2, imm=31)]
    for off in (0, 1, 2, 3, 7, 14, 16, 31, 100):
        code += [E(O.LB, 6, 5, imm=off), E(O.LBU, 7, 5, imm=off)]
    for off in (0, 2, 6, 12, 16, 30):
        code += [E(O.LH, 6, 5, imm=off), E(O.LHU, 7, 5, imm=off)]
    for off in (0, 4, 8, 16, 28):
        code += [E(O.LW, 8, 5, imm=off)]
    code += [E(O.LD, 9, 5, imm=0), E(O.LD, 9, 5, imm=16), E(O.LD, 9, 5, imm=1024)]
    # Q1/Q6: sign-extended 64-bit register values in raw compares, DIV on "negative" raw values, MULH on raw operands
    code += [E(O.LB, 6, 5, imm=31), E(O.LB, 7, 5, imm=31), E(O.SEQ, 8, 6, 7), E(O.ADDI, 7, 7, imm=0), E(O.SEQ, 8, 6, 7), E(O.SNE, 9, 6, 7),
             spec.beq(6, 7, 8), A(12, 0, 1), A(4, 0, 3), E(O.DIV, 13, 6, 4), E(O.REM, 14, 6, 4), E(O.MULH, 15, 6, 6), E(O.CMOVNZ, 12, 6, 6),
             E(O.SD, rs1=5, rs2=6, imm=40), E(O.LD, 3, 5, imm=40), E(O.ADD, 3, 3, 0)]
    return _p(code + [EB]), [], {}


def jumps_and_links():
    code = [spec.jal(1, 8), EB, A(2, 1, 0), E(O.JALR, 3, 1, imm=12 + 1),    # jalr target = (r1 + 13) & ~1
            EB, EB, A(4, 3, 0), spec.jal(0, 8), EB, A(5, 0, 7), E(O.BLT, rs1=0, rs2=5, imm=8), EB, E(O.BGEU, rs1=5, rs2=0, imm=8), EB,
            A(6, 0, -1), E(O.BLT, rs1=6, rs2=0, imm=8), EB, E(O.BLTU, rs1=6, rs2=0, imm=8), E(O.BGE, rs1=0, rs2=6, imm=8), EB, EB]
    return _p(code), [], {}


def q9_access_at_own_pc():
    """Q9: any data op whose address equals the fetching pc is dropped from the row (vm.rs:295)."""
    code = [A(1, 0, 0x1000 + 8), A(2, 0, 0), spec.lw(3, 1, 0), A(1, 0, 0x1000 + 16), spec.sw(1, 3, 0), EB]
    # instr[2] at pc 0x1008 loads from 0x1008; instr[4] at pc 0x1010 stores to 0x1010 (overwrites itself after the fetch)
    return _p(code), [], {}


def self_modifying():
    """strict protection is off at run time (vm.rs:175): a store can rewrite a later instruction."""
    target = 0x1000 + 4 * 6
    new_word = A(7, 0, 1234)
    code = li40(1, new_word) + [A(2, 0, target)]         # 5 + 1 = 6 instrs
    code = li40(1, new_word)[:5] + [A(2, 0, 0x1000 + 4 * 8), spec.sw(2, 1, 0), A(0, 0, 0), EB, EB]
    # word index 8 (an EBREAK) is replaced by `addi r7, r0, 1234` before it is fetched
    return _p(code), [], {}


def cmov_false_keeps_bound():
    code = [A(1, 0, 5), A(2, 0, 9), E(O.CMOV, 2, 1, 0), E(O.CMOVZ, 2, 1, 1), E(O.CMOVNZ, 2, 1, 1), E(O.CMOVZ, 3, 1, 0), EB]
    return _p(code), [], {}


# ---- syscalls --------------------------------------------------------------------------------------
def _store_bytes(base_reg, data: bytes):
    out = []
    for i, b in enumerate(data):
        out += [A(9, 0, b), E(O.SB, rs1=base_reg, rs2=9, imm=i)]
    return out


def sha256_hello():                 # syscall.rs:280-318
    code = [A(5, 0, 0x2000)] + _store_bytes(5, b"hello") + [A(11, 5, 0), A(12, 0, 5), A(13, 0, 0x3000), A(10, 0, 3), EC,
                                                          spec.lw(1, 13, 0), spec.lw(2, 13, 28)] + _exit0()
    return _p(code), [], {}


def hashes_all():
    msg = bytes((i * 7 + 3) & 0xFF for i in range(70))
    code = [A(5, 0, 0x2000)] + _store_bytes(5, msg)
    for num, ln, out in ((3, 0, 0x3000), (3, 55, 0x3020), (3, 56, 0x3040), (3, 70, 0x3060), (5, 0, 0x3080), (5, 70, 0x30A0), (6, 0, 0x30C0), (6, 70, 0x30E0)):
        code += [A(11, 5, 0), A(12, 0, ln), A(13, 0, out), A(10, 0, num), EC, A(1, 14, 0)]
    # unaligned input pointer is fine (byte reads); chain: hash the previous digest
    code += [A(11, 0, 0x3001), A(12, 0, 31), A(13, 0, 0x3100), A(10, 0, 3), EC, A(11, 0, 0x3100), A(12, 0, 32), A(13, 0, 0x3120), A(10, 0, 5), EC]
    return _p(code + _exit0()), [], {}


def blake3_multi_chunk():
    """Input > 1024 bytes exercises BLAKE3's chunk tree; zero-filled memory reads as zeros (memory.rs:301-305)."""
    code = [A(5, 0, 0x4000), A(9, 0, 0xAB), E(O.SB, rs1=5, rs2=9, imm=1500)]
    for ln, out in ((1024, 0x3000), (1025, 0x3020), (2048, 0x3040), (2049, 0x3060), (3073, 0x3080), (5000, 0x30A0)):
        code += [A(11, 5, 0), A(12, 0, ln), A(13, 0, out), A(10, 0, 6), EC]
    code += [A(11, 5, 0), A(12, 0, 137), A(13, 0, 0x3100), A(10, 0, 5), EC, A(11, 5, 0), A(12, 0, 300), A(13, 0, 0x3120), A(10, 0, 5), EC]
    return _p(code + _exit0()), [], {}


def sha_chain_small():
    return spec.sha256_chain_program().to_bytes(), [], {"max_cycles": 600}


# ---- deferred carry model (execute.rs:888-1003) -----------------------------------------------------
def deferred_add_branch():          # deferred_integration_test.rs:21-98
    code = [A(1, 0, 100), A(2, 0, 200), spec.add(3, 1, 2), A(4, 0, 300), spec.beq(3, 4, 8), A(5, 0, 1), A(5, 0, 42)] + _exit0()
    return _p(code), [], {"enable_deferred_model": True}


def deferred_chain_store():         # deferred_integration_test.rs:101-160
    code = [A(1, 0, 10), A(1, 1, 20), A(1, 1, 30), A(1, 1, 40), A(1, 1, 50), E(O.SW, rs1=0, rs2=1, imm=0x10000)] + _exit0()
    return _p(code), [], {"enable_deferred_model": True}


def deferred_add_sub_mix():         # deferred_integration_test.rs:163-224
    code = [A(1, 0, 100), A(2, 0, 50), spec.add(3, 1, 2), A(4, 0, 30), spec.sub(5, 3, 4), E(O.ANDI, 6, 5, imm=0xFFFF)] + _exit0()
    return _p(code), [], {"enable_deferred_model": True}


def deferred_negative_and_overflow():
    """Two's-complement ADDI (normalize.rs:331-360), SUB underflow wrap (deferred.rs:184-187), limb overflow
    forcing source normalization (deferred.rs:100-113), non-deferred writers leaving the Accumulated flag set (Q10)."""
    code = [A(1, 0, 32767), A(1, 1, 1), A(2, 1, -16), A(3, 0, 5), spec.sub(4, 3, 1), spec.sub(4, 4, 1), E(O.XOR, 4, 4, 3), spec.add(4, 4, 4)]
    code += [A(6, 0, 65535), E(O.SLLI, 6, 6, imm=4), A(6, 6, 0)]
    code += [spec.add(6, 6, 6)] * 14                    # limb0 doubles until it would exceed 2^30
    code += [A(7, 6, -1)] + [spec.add(7, 7, 6)] * 3 + [A(8, 0, 1), A(10, 0, 1), EC, spec.add(9, 10, 10), E(O.MUL, 9, 9, 7),
                                                       E(O.SLTU, 11, 7, 6), E(O.SD, rs1=0, rs2=7, imm=0x4000), E(O.SRAI, 12, 7, imm=3), spec.bne(7, 6, 8), EB, EB]
    return _p(code), [777], {"enable_deferred_model": True}


def deferred_fib():
    return spec.fib_program(40).to_bytes(), [], {"enable_deferred_model": True, "enable_range_checking": True}


def fib_rc():
    """fib with range checking: every ADD whose bound exceeds 40 bits defers and is flushed at the next BNE."""
    return spec.fib_program(200).to_bytes(), [], {"enable_range_checking": True}


ALL = {f.__name__: f for f in (fib_rc,
    exit_42, io_echo, echo5, jal_self_cycle_limit, basic_add_ebreak, mem_sw_lw, timestamps, beq_skip, sum_1_to_5, fib30, rc_doubling,
    rc_small_consts, rc_many_pending, rc_config_30bit, alu_all, loads_stores, jumps_and_links, q9_access_at_own_pc, self_modifying,
    cmov_false_keeps_bound, sha256_hello, hashes_all, blake3_multi_chunk, sha_chain_small, deferred_add_branch, deferred_chain_store,
    deferred_add_sub_mix, deferred_negative_and_overflow, deferred_fib)}


# ---- programs that must fail with a specific RuntimeError (error.rs:7-37) ---------------------------
def err_div_zero(): return _p([A(1, 0, 5), E(O.DIV, 2, 1, 0), EB]), [], {}, 3
def err_remu_zero(): return _p([A(1, 0, 5), E(O.REMU, 2, 1, 0), EB]), [], {}, 3
def err_misaligned_lw(): return _p([A(1, 0, 0x2002), spec.lw(2, 1, 0), EB]), [], {}, 1
def err_misaligned_sd(): return _p([A(1, 0, 0x2004), E(O.SD, rs1=1, rs2=0, imm=0), EB]), [], {}, 1
def err_misaligned_sha_out(): return _p([A(11, 0, 0x2000), A(12, 0, 0), A(13, 0, 0x3002), A(10, 0, 3), EC, EB]), [], {}, 1
def err_invalid_syscall(): return _p([A(10, 0, 999), EC]), [], {}, 4
def err_poseidon2(): return _p([A(10, 0, 4), EC]), [], {}, 6          # crypto.rs:306-315
def err_unknown_opcode(): return _p([A(1, 0, 1), 0x0000007F]), [], {}, 5
def err_misaligned_pc(): return _p([A(1, 0, 0x1000 + 10), E(O.JALR, 0, 1, imm=0), EB]), [], {}, 6    # target & ~1 = 0x100a
def err_debug_format():
    p = Program.from_code([EB]); p.header.entry_point = 32
    return p.to_bytes(), [], {}, 7
def err_bad_magic(): return b"\x00" * 40, [], {}, 7
def err_truncated():
    return _p([EB, EB])[:-3], [], {}, 7


ERRORS = {f.__name__: f for f in (err_div_zero, err_remu_zero, err_misaligned_lw, err_misaligned_sd, err_misaligned_sha_out,
                                  err_invalid_syscall, err_poseidon2, err_unknown_opcode, err_misaligned_pc, err_debug_format,
                                  err_bad_magic, err_truncated)}


# ---- seeded random programs --------------------------------------------------------------------------
_R_OPS = [O.ADD, O.SUB, O.MUL, O.MULH, O.AND, O.OR, O.XOR, O.SLL, O.SRL, O.SRA, O.SLTU, O.SGEU, O.SLT, O.SGE, O.SEQ, O.SNE, O.CMOV, O.CMOVZ, O.CMOVNZ]
_I_OPS = [O.ADDI, O.ANDI, O.ORI, O.XORI]
_GP = [1, 2, 3, 4, 7, 8, 9, 14, 15]       # r5 = memory base, r6 = non-zero divisor, r10-r13 = syscall scratch


def random_program(seed: int, n_instr: int = 300, range_checking: bool = False, hashes: bool = True, wide_safe: bool = False):
    """`wide_safe` (proof mode 4): the five wide opcodes read operands cut below 2^40 first (SRLI by >= 24 into the syscall scratch r12 / r13, or r0 / r6), which is
    what mode 4 states them on; MULH leaves the generic R-type pool for the same reason.  Zero dividends, zero factors and equal operands stay in the draw."""
    rng = np.random.default_rng(seed)
    ri = lambda lo, hi: int(rng.integers(lo, hi))  # noqa: E731
    reg = lambda: _GP[ri(0, len(_GP))]              # noqa: E731
    src = lambda: ([0] + _GP + [5, 6])[ri(0, len(_GP) + 3)]  # noqa: E731
    code = [A(5, 0, 0x8000), E(O.SLLI, 5, 5, imm=1)] + li40(6, int(rng.integers(1, 1 << 40)) | 1)
    for r in _GP:
        code += li40(r, int(rng.integers(0, 1 << 40))) if rng.random() < 0.6 else [A(r, 0, ri(-65536, 65536))]
    body = []
    while len(body) < n_instr:
        k = rng.random()
        if k < 0.40:
            op = _R_OPS[ri(0, len(_R_OPS))]
            if wide_safe and op == O.MULH:
                op = O.MUL
            body.append(E(op, reg(), src(), src()))
        elif k < 0.55:
            body.append(E(_I_OPS[ri(0, len(_I_OPS))], reg(), src(), imm=ri(-65536, 65536)))
        elif k < 0.63:
            body.append(E([O.SLLI, O.SRLI, O.SRAI][ri(0, 3)], reg(), src(), imm=ri(0, 64) if rng.random() < 0.9 else ri(64, 256)))
        elif k < 0.73:
            w = [1, 2, 4, 8][ri(0, 4)]
            off = ri(0, 256) * w
            ops = {1: [O.LB, O.LBU], 2: [O.LH, O.LHU], 4: [O.LW], 8: [O.LD]}[w]
            body.append(E(ops[ri(0, len(ops))], reg(), 5, imm=off))
        elif k < 0.82:
            w = [1, 2, 4, 8][ri(0, 4)]
            body.append(E({1: O.SB, 2: O.SH, 4: O.SW, 8: O.SD}[w], rs1=5, rs2=src(), imm=ri(0, 256) * w))
        elif k < 0.86 and wide_safe:
            body += [E(O.SRLI, 12, src(), imm=ri(24, 64)), E(O.SRLI, 13, src(), imm=ri(24, 64)), E(O.ORI, 13, 13, imm=1)]      # r13 != 0: a zero divisor is a VM error
            op = [O.DIVU, O.REMU, O.DIV, O.REM, O.MULH][ri(0, 5)]
            b = [13, 12, 6, 0][ri(0, 4)] if op == O.MULH else [13, 13, 6][ri(0, 3)]
            body.append(E(op, reg(), [12, 12, 12, 13, 6, 0][ri(0, 6)], b))
        elif k < 0.86:
            body.append(E([O.DIVU, O.REMU, O.DIV, O.REM][ri(0, 4)], reg(), src(), 6 if rng.random() < 0.97 else src()))
        elif k < 0.93:
            body.append(E([O.BEQ, O.BNE, O.BLT, O.BGE, O.BLTU, O.BGEU][ri(0, 6)], rs1=src(), rs2=src(), imm=4 * ri(1, 5)))
        elif k < 0.94:
            body.append(spec.jal(reg() if rng.random() < 0.5 else 0, 4 * ri(1, 4)))
        elif k < 0.95:                                               # a computed jump: link the next address, then JALR past 1..3 instructions (even and odd sums)
            body += [spec.jal(7, 4), E(O.JALR, reg() if rng.random() < 0.5 else 0, 7, imm=4 * ri(1, 5) + ri(0, 2))]
        elif k < 0.97:
            body += [A(11, src(), 0), A(10, 0, 2), EC] if rng.random() < 0.5 else [A(10, 0, 1), EC, A(reg(), 10, 0)]
        elif hashes:
            num = [3, 5, 6][ri(0, 3)]
            body += [A(11, 5, ri(0, 512)), A(12, 0, ri(0, 200)), A(13, 5, 1024 + 4 * ri(0, 128)), A(10, 0, num), EC, A(reg(), 14, 0)]
    code += body + [EB] * 6
    inputs = [int(x) for x in rng.integers(0, 1 << 62, size=5)]
    return _p(code), inputs


def off_code(name):
    """The program `name` with its data moved off the code segment (mem_sw_lw, timestamps, rc_doubling: the reference's tests store at 0x1000); every other program as it is."""
    f = globals()[name]
    import inspect
    return f(base=OFF_CODE) if "base" in inspect.signature(f).parameters else f()
