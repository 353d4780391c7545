"""ZKIR v3.4 input contract, host side: opcodes, 32-bit encodings and the program blob.

Mirrors (names and meaning) the reference's spec/encoder surface that feeds the hot path:
  * Opcode                 zkir-spec/src/opcode.rs:24-144
  * encode(...)            zkir-assembler/src/encoder.rs:18-151 (bit layout zkir-spec/src/encoding.rs:23-60)
  * Config                 zkir-spec/src/config.rs:10-56
  * ProgramHeader/Program  zkir-spec/src/program.rs:62-346 (32-byte LE header + code words + data)

This is plain host logic (no device work): it only builds the byte blob that `zkir_exec`
(include/zkir_amd.h) consumes.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Sequence

MAGIC = 0x52494B5A      # program.rs:37  ("ZKIR" little-endian)
VERSION = 0x00030004    # program.rs:40
CODE_BASE = 0x1000      # vm.rs:155
NUM_REGISTERS = 16      # register.rs:10


class Opcode(IntEnum):
    """opcode.rs:24-144 (7-bit opcodes)."""
    ADD = 0x00; SUB = 0x01; MUL = 0x02; MULH = 0x03; DIVU = 0x04; REMU = 0x05; DIV = 0x06; REM = 0x07; ADDI = 0x08
    AND = 0x10; OR = 0x11; XOR = 0x12; ANDI = 0x13; ORI = 0x14; XORI = 0x15
    SLL = 0x18; SRL = 0x19; SRA = 0x1A; SLLI = 0x1B; SRLI = 0x1C; SRAI = 0x1D
    SLTU = 0x20; SGEU = 0x21; SLT = 0x22; SGE = 0x23; SEQ = 0x24; SNE = 0x25
    CMOV = 0x26; CMOVZ = 0x27; CMOVNZ = 0x28
    LB = 0x30; LBU = 0x31; LH = 0x32; LHU = 0x33; LW = 0x34; LD = 0x35
    SB = 0x38; SH = 0x39; SW = 0x3A; SD = 0x3B
    BEQ = 0x40; BNE = 0x41; BLT = 0x42; BGE = 0x43; BLTU = 0x44; BGEU = 0x45
    JAL = 0x48; JALR = 0x49
    ECALL = 0x50; EBREAK = 0x51


R_TYPE = {Opcode.ADD, Opcode.SUB, Opcode.MUL, Opcode.MULH, Opcode.DIVU, Opcode.REMU, Opcode.DIV, Opcode.REM,
          Opcode.AND, Opcode.OR, Opcode.XOR, Opcode.SLL, Opcode.SRL, Opcode.SRA,
          Opcode.SLTU, Opcode.SGEU, Opcode.SLT, Opcode.SGE, Opcode.SEQ, Opcode.SNE,
          Opcode.CMOV, Opcode.CMOVZ, Opcode.CMOVNZ}
I_TYPE = {Opcode.ADDI, Opcode.ANDI, Opcode.ORI, Opcode.XORI, Opcode.LB, Opcode.LBU, Opcode.LH, Opcode.LHU,
          Opcode.LW, Opcode.LD, Opcode.JALR}
SHIFT_I = {Opcode.SLLI, Opcode.SRLI, Opcode.SRAI}
S_TYPE = {Opcode.SB, Opcode.SH, Opcode.SW, Opcode.SD}
B_TYPE = {Opcode.BEQ, Opcode.BNE, Opcode.BLT, Opcode.BGE, Opcode.BLTU, Opcode.BGEU}


def _r(op: int, rd: int, rs1: int, rs2: int, funct: int = 0) -> int:      # encoder.rs:100-108
    return (op & 0x7F) | ((rd & 0xF) << 7) | ((rs1 & 0xF) << 11) | ((rs2 & 0xF) << 15) | ((funct & 0x1FFF) << 19)


def _i(op: int, rd: int, rs1: int, imm: int) -> int:                       # encoder.rs:112-119 (imm silently masked to 17 bits)
    return (op & 0x7F) | ((rd & 0xF) << 7) | ((rs1 & 0xF) << 11) | ((imm & 0x1FFFF) << 15)


def _j(op: int, rd: int, offset: int) -> int:                              # encoder.rs:145-151
    return (op & 0x7F) | ((rd & 0xF) << 7) | ((offset & 0x1FFFFF) << 11)


def encode(op: Opcode, rd: int = 0, rs1: int = 0, rs2: int = 0, imm: int = 0) -> int:
    """encode(&Instruction) of zkir-assembler/src/encoder.rs:18-96.

    Field meaning per format: R-type (rd, rs1, rs2); I-type (rd, rs1, imm); shift-immediate
    (rd, rs1, imm=shamt); S-type and B-type (rs1, rs2, imm/offset) with rs1 in bits 10:7 and rs2 in
    bits 14:11; JAL (rd, imm=offset)."""
    op = Opcode(op)
    if op in R_TYPE:
        return _r(op, rd, rs1, rs2)
    if op in I_TYPE:
        return _i(op, rd, rs1, imm)
    if op in SHIFT_I:
        return _i(op, rd, rs1, imm & 0xFF)
    if op in S_TYPE or op in B_TYPE:
        return _i(op, rs1, rs2, imm)
    if op == Opcode.JAL:
        return _j(op, rd, imm)
    return _i(op, 0, 0, 0)                                                  # ECALL / EBREAK


# ----- convenience constructors in the reference's `Instruction::X { .. }` vocabulary -------------
def add(rd, rs1, rs2): return encode(Opcode.ADD, rd, rs1, rs2)
def sub(rd, rs1, rs2): return encode(Opcode.SUB, rd, rs1, rs2)
def mul(rd, rs1, rs2): return encode(Opcode.MUL, rd, rs1, rs2)
def addi(rd, rs1, imm): return encode(Opcode.ADDI, rd, rs1, imm=imm)
def slli(rd, rs1, shamt): return encode(Opcode.SLLI, rd, rs1, imm=shamt)
def lw(rd, rs1, imm): return encode(Opcode.LW, rd, rs1, imm=imm)
def sw(rs1, rs2, imm): return encode(Opcode.SW, rs1=rs1, rs2=rs2, imm=imm)   # mem[rs1+imm] = rs2
def beq(rs1, rs2, off): return encode(Opcode.BEQ, rs1=rs1, rs2=rs2, imm=off)
def bne(rs1, rs2, off): return encode(Opcode.BNE, rs1=rs1, rs2=rs2, imm=off)
def jal(rd, off): return encode(Opcode.JAL, rd, imm=off)
def ecall(): return encode(Opcode.ECALL)
def ebreak(): return encode(Opcode.EBREAK)


@dataclass
class Config:
    """config.rs:10-56."""
    limb_bits: int = 20
    data_limbs: int = 2
    addr_limbs: int = 2

    def validate(self) -> None:                                             # config.rs:154-174
        if not (16 <= self.limb_bits <= 30):
            raise ValueError("InvalidLimbBits")
        if self.limb_bits % 2:
            raise ValueError("OddLimbBits")
        if not (1 <= self.data_limbs <= 4):
            raise ValueError("InvalidDataLimbs")
        if not (1 <= self.addr_limbs <= 2):
            raise ValueError("InvalidAddrLimbs")

    def data_bits(self) -> int:
        return self.limb_bits * self.data_limbs


@dataclass
class ProgramHeader:
    """program.rs:62-214 (32 bytes, little-endian)."""
    magic: int = MAGIC
    version: int = VERSION
    limb_bits: int = 20
    data_limbs: int = 2
    addr_limbs: int = 2
    flags: int = 0
    entry_point: int = CODE_BASE
    code_size: int = 0
    data_size: int = 0
    bss_size: int = 0
    stack_size: int = 1 << 20

    SIZE = 32

    def to_bytes(self) -> bytes:                                            # program.rs:170-186
        return struct.pack("<IIBBBBIIIII", self.magic, self.version, self.limb_bits, self.data_limbs, self.addr_limbs,
                           self.flags, self.entry_point, self.code_size, self.data_size, self.bss_size, self.stack_size)

    @classmethod
    def from_bytes(cls, b: bytes) -> "ProgramHeader":                       # program.rs:189-213
        if len(b) < cls.SIZE:
            raise ValueError(f"Invalid header size: expected 32 bytes, found {len(b)} bytes")
        h = cls(*struct.unpack("<IIBBBBIIIII", b[:32]))
        h.validate()
        return h

    def validate(self) -> None:                                             # program.rs:147-167
        if self.magic != MAGIC:
            raise ValueError(f"Invalid program magic: expected 0x5A4B4952, got {self.magic:#010x}")
        if self.version != VERSION:
            raise ValueError(f"Invalid program version: expected {VERSION:#010x}, found {self.version:#010x}")
        Config(self.limb_bits, self.data_limbs, self.addr_limbs).validate()


@dataclass
class Program:
    """program.rs:240-346."""
    header: ProgramHeader = field(default_factory=ProgramHeader)
    code: List[int] = field(default_factory=list)
    data: bytes = b""

    @classmethod
    def from_code(cls, code: Sequence[int], data: bytes = b"", config: Config | None = None) -> "Program":
        """The `create_program_from_instructions` helper pattern of the reference's tests (vm.rs:419-431),
        additionally keeping header.data_size consistent so the blob passes Program::from_bytes."""
        p = cls()
        if config is not None:
            config.validate()
            p.header.limb_bits, p.header.data_limbs, p.header.addr_limbs = config.limb_bits, config.data_limbs, config.addr_limbs
        p.code = [c & 0xFFFFFFFF for c in code]
        p.data = bytes(data)
        p.header.code_size = 4 * len(p.code)
        p.header.data_size = len(p.data)
        return p

    def to_bytes(self) -> bytes:                                            # program.rs:300-315
        return self.header.to_bytes() + b"".join(struct.pack("<I", w) for w in self.code) + self.data

    @classmethod
    def from_bytes(cls, b: bytes) -> "Program":                             # program.rs:318-346
        h = ProgramHeader.from_bytes(b)
        code_end = 32 + h.code_size
        data_end = code_end + h.data_size
        if len(b) < data_end:
            raise ValueError(f"Invalid program size: expected {data_end} bytes, found {len(b)} bytes")
        n = h.code_size // 4
        code = list(struct.unpack(f"<{n}I", b[32:32 + 4 * n]))
        if 4 * len(code) != h.code_size:
            raise ValueError(f"Invalid code size: expected {h.code_size} bytes, found {4 * len(code)} bytes")
        return cls(h, code, bytes(b[code_end:data_end]))


# ----- workloads named by BASELINE.json / SURVEY.md §8(d) ----------------------------------------
def fib_program(n: int) -> Program:
    """v3.4 Fibonacci of tests/cross_module.rs:145-164 generalised: writes fib(n) then exits 0.
    n >= 2; the loop counter n-1 must fit the 17-bit immediate (Q11), i.e. n <= 65536."""
    assert 2 <= n <= 65536
    return Program.from_code([
        addi(1, 0, 0), addi(2, 0, 1), addi(3, 0, n - 1),
        add(4, 1, 2), addi(1, 2, 0), addi(2, 4, 0), addi(3, 3, -1), bne(3, 0, -16),
        addi(11, 2, 0), addi(10, 0, 2), ecall(),
        addi(10, 0, 0), addi(11, 0, 0), ecall(),
    ])


def fib_endless_program() -> Program:
    """Same loop body, never exits: run with max_cycles = 2^k for exactly 2^k rows (halt = CycleLimit,
    vm.rs:211-214).  The counter r3 just decrements mod 2^40 (first zero after 2^40 iterations)."""
    return Program.from_code([
        addi(1, 0, 0), addi(2, 0, 1), addi(3, 0, 0),
        add(4, 1, 2), addi(1, 2, 0), addi(2, 4, 0), addi(3, 3, -1), bne(3, 0, -16),
        jal(0, -20),
    ])


def compare_loop_program() -> Program:
    """An endless loop over every opcode family the AIR (v3) constrains besides the fib loop's: SUB (with and without a borrow out of
    40 bits), SLTU / SGEU, SEQ / SNE, BEQ / BNE, BLTU / BGEU — every comparison comes out both ways and every branch is both taken and
    not taken (a runs after a doubling b and passes a threshold; the first 600 cycles see all of it).  Run with max_cycles (halt = CycleLimit)."""
    def r(op, rd, rs1, rs2): return encode(op, rd, rs1, rs2)
    def b(op, rs1, rs2, off): return encode(op, rs1=rs1, rs2=rs2, imm=off)
    O = Opcode
    return Program.from_code([
        addi(1, 0, 0), addi(2, 0, 7), addi(3, 0, 100), addi(10, 0, 1), addi(11, 0, 500),    # i, a, b, the previous iteration's (a < 500), the threshold
        # L (pc 0x1014):
        sub(4, 3, 2), sub(5, 2, 3),                              # b - a, a - b: one of the two wraps below zero (mod 2^40)
        r(O.SLTU, 6, 2, 11), r(O.SGEU, 7, 2, 11),                # a < 500, a >= 500
        r(O.SEQ, 8, 6, 10), r(O.SNE, 9, 6, 10),                  # same outcome as last time?
        add(2, 2, 6), addi(2, 2, 45),                            # a += 45 + (a < 500)
        b(O.BLTU, 2, 3, 8),                                      # a < b: skip the next instruction
        add(3, 3, 3),                                            # b *= 2: a needs longer and longer to catch up
        b(O.BGEU, 2, 11, 8),                                     # a >= 500: skip
        sub(3, 3, 9),                                            # b -= (outcome changed)
        b(O.BEQ, 6, 10, 8),                                      # outcome repeated: skip the count
        addi(1, 1, 1),                                           # i += 1
        addi(10, 6, 0),
        b(O.BNE, 1, 0, -60),                                     # i != 0: back to L
        jal(0, -64),                                             # i still 0: back to L as well
    ])


def call_loop_program() -> Program:
    """An endless loop around a subroutine call: JAL links, the callee returns through JALR — once with an even target, once with an odd one
    and a negative immediate (the cleared bit) — with signed branches (BLT / BGE: class "other, jumps" of the AIR, v4) both ways and
    MUL / SLLI rows in between (class "other": sequential).  Run with max_cycles."""
    def b(op, rs1, rs2, off): return encode(op, rs1=rs1, rs2=rs2, imm=off)
    O = Opcode
    return Program.from_code([
        addi(1, 0, 0), addi(2, 0, 5), addi(6, 0, 20),            # i, x, a bound
        # L (0x100C):
        jal(15, 24),                                             # call F (0x1024), link in r15
        addi(1, 1, 1),                                           # i += 1
        b(O.BLT, 1, 2, 8),                                       # i < x (signed): skip the next instruction
        addi(2, 2, 7),                                           # x += 7
        b(O.BGE, 1, 6, -16),                                     # i >= 20: back to L this way
        jal(0, -20),                                             # else that way
        # F (0x1024):
        mul(3, 1, 2), slli(4, 3, 2),                             # "other" rows: pc + 4
        addi(14, 15, 5),                                         # return address + 5
        encode(O.SEQ, 5, 4, 0),                                  # i * x * 4 == 0 ?  (the first call only)
        b(O.BEQ, 5, 0, 8),                                       # no: return the odd way
        encode(O.JALR, 13, 15, imm=0),                           # return to r15, link r13
        encode(O.JALR, 0, 14, imm=-4),                           # return to (r15 + 5 - 4) & ~1 = r15: negative immediate, odd sum
    ])


def signed_loop_program() -> Program:
    """An endless loop over the SIGNED comparisons (AIR v5): SLT / SGE and BLT / BGE on a counter that walks from -6 to 8 and back, against a
    negative and a positive threshold, the most negative value (2^39) and the largest positive one (2^39 - 1); SLTU / BLTU on the same
    operands for contrast (a negative value is a huge unsigned one).  Every comparison comes out both ways, every branch is both taken and
    not taken.  Run with max_cycles (halt = CycleLimit)."""
    def r(op, rd, rs1, rs2): return encode(op, rd, rs1, rs2)
    def b(op, rs1, rs2, off): return encode(op, rs1=rs1, rs2=rs2, imm=off)
    O = Opcode
    return Program.from_code([
        addi(1, 0, -6), addi(2, 0, -2), addi(3, 0, 3), addi(13, 0, 8),       # i, a negative and a positive threshold, the loop bound
        addi(15, 0, 1), slli(15, 15, 39), addi(14, 15, -1),                  # the most negative value 2^39 and the largest positive one 2^39 - 1
        # L (pc 0x101C):
        r(O.SLT, 4, 1, 2), r(O.SGE, 5, 1, 2),                                # i < -2, i >= -2
        r(O.SLT, 6, 3, 1), r(O.SGE, 7, 3, 1),                                # 3 < i, 3 >= i
        r(O.SLTU, 8, 1, 3), r(O.SGEU, 9, 1, 3),                              # the same operands unsigned: a negative i is huge
        r(O.SLT, 10, 15, 1), r(O.SGE, 11, 14, 1),                            # min < i (true unless i = min ... always true here), max >= i (always)
        r(O.SLT, 12, 14, 15),                                                # max < min: never
        b(O.BLT, 1, 0, 8),                                                   # i < 0: skip the next instruction
        addi(4, 4, 16),
        b(O.BGE, 1, 3, 8),                                                   # i >= 3: skip
        addi(5, 5, 16),
        b(O.BLT, 2, 1, 8),                                                   # -2 < i: skip
        addi(6, 6, 16),
        b(O.BGE, 2, 1, 8),                                                   # -2 >= i: skip
        addi(7, 7, 16),
        b(O.BLTU, 1, 3, 8),                                                  # i < 3 unsigned: only for i = 0, 1, 2
        addi(8, 8, 16),
        b(O.BLT, 14, 15, 8),                                                 # max < min: never taken
        addi(1, 1, 1),                                                       # i += 1
        b(O.BLT, 1, 13, -84),                                                # i < 8: back to L
        addi(1, 0, -6),                                                      # else start over from -6
        b(O.BGE, 14, 15, -92),                                               # max >= min: always taken, back to L
    ])


def cmov_loop_program() -> Program:
    """An endless loop over the conditional moves (AIR v6): CMOV / CMOVNZ (move if rs2 != 0) and CMOVZ (move if rs2 == 0) with the condition
    toggling every iteration, a condition that is non-zero only above bit 20, a raw 64-bit source and condition (a sign-extended byte load:
    bits above 40 set, Q1), and rd = r0.  Run with max_cycles (halt = CycleLimit)."""
    def r(op, rd, rs1, rs2): return encode(op, rd, rs1, rs2)
    O = Opcode
    return Program.from_code([
        addi(1, 0, 0), addi(2, 0, 7), addi(13, 0, 1), addi(10, 0, 0),         # i, a value to move, the constant 1, the toggling condition
        addi(5, 0, 0x80), addi(6, 0, 0x8000), slli(6, 6, 1), sw(6, 5, 0),     # mem[0x10000] = 0x80
        encode(O.LB, 7, 6, imm=0),                                            # r7 = 0xFFFF_FFFF_FFFF_FF80: all three limbs non-zero (execute.rs:477-487)
        addi(9, 0, 1), slli(9, 9, 20),                                        # r9 = 2^20: non-zero only in the second limb
        # L (pc 0x102C):
        r(O.CMOV, 11, 2, 10), r(O.CMOVZ, 12, 7, 10), r(O.CMOVNZ, 14, 9, 10),  # move iff the toggle is 1 / 0 / 1
        r(O.CMOV, 0, 2, 13),                                                  # condition true, rd = r0: nothing is written
        r(O.CMOVZ, 15, 2, 9),                                                 # condition 2^20 != 0: not moved
        r(O.CMOVNZ, 15, 7, 7),                                                # a raw 64-bit condition and source: moved with its bits above 40
        r(O.CMOVZ, 4, 7, 0),                                                  # condition r0 == 0: always moved
        addi(11, 0, 0), addi(12, 0, 0), addi(14, 0, 0), addi(15, 0, 0), addi(4, 0, 0),
        sub(10, 13, 10),                                                      # toggle
        addi(1, 1, 1),
        jal(0, -56),                                                          # back to L
    ])


def memory_loop_program(n: int) -> Program:
    """A loop over an array of n 8-byte cells at 0x10000 (AIR mode 3: the memory argument): cell i <- 3 i (SD), then read back as a word, an unsigned halfword at offset 2
    and a signed byte (LW / LHU / LB: three windows of the cell just written), summed; after n iterations (13 rows each, 4 of them memory accesses) the sum is
    WRITTEN and the run exits 0.  Run it whole, or with max_cycles for a prefix (halt = CycleLimit)."""
    assert 1 <= n <= 65535
    return Program.from_code([
        addi(6, 0, 0x8000), slli(6, 6, 1), addi(1, 0, 0), addi(3, 0, n), addi(4, 0, 0),
        # L:
        add(2, 1, 1), add(2, 2, 1), encode(Opcode.SD, rs1=6, rs2=2, imm=0),
        lw(7, 6, 0), encode(Opcode.LHU, 8, 6, imm=2), encode(Opcode.LB, 9, 6, imm=0),
        add(4, 4, 7), add(4, 4, 8), add(4, 4, 9),
        addi(6, 6, 8), addi(1, 1, 1), addi(3, 3, -1), bne(3, 0, -48),
        addi(11, 4, 0), addi(10, 0, 2), ecall(), addi(10, 0, 0), addi(11, 0, 0), ecall(),
    ])


def wide_loop_program() -> Program:
    """An ENDLESS loop over the five wide-arithmetic opcodes (AIR mode 4: MULH DIVU REMU DIV REM on operands below 2^40; run with max_cycles, halt = CycleLimit): a 40-bit
    state x <- x * 0x2545F + 0x14057B (MUL / ADDI: wraps mod 2^40), a divisor d = (x >> 17) | 1, then MULH(x, x), DIVU, REMU, DIV, REM of x by d and a MULH of the quotient by
    the divisor, accumulated with XOR into a checksum that is stored and read back (one SD / LD per iteration): 18 rows per iteration, 6 of them wide, every operand 40 bits."""
    E, O = encode, Opcode
    return Program.from_code([
        addi(1, 0, 12345), addi(2, 0, 0x2545), slli(2, 2, 4), addi(2, 2, 0xF), addi(12, 0, 0), addi(6, 0, 0x8000), slli(6, 6, 1),      # x, the multiplier 0x2545F, checksum, base 0x10000
        # L:
        mul(1, 1, 2), addi(1, 1, 0x4057), E(O.SRLI, 3, 1, imm=17), E(O.ORI, 3, 3, imm=1),
        E(O.MULH, 4, 1, 1), E(O.DIVU, 5, 1, 3), E(O.REMU, 7, 1, 3), E(O.DIV, 8, 4, 3), E(O.REM, 9, 4, 3), E(O.MULH, 10, 5, 3),
        E(O.XOR, 12, 12, 4), E(O.XOR, 12, 12, 5), E(O.XOR, 12, 12, 7), E(O.XOR, 12, 12, 8), E(O.XOR, 12, 12, 9), E(O.XOR, 12, 12, 10),
        E(O.SD, rs1=6, rs2=12, imm=0), E(O.LD, 13, 6, imm=0), jal(0, -72),
    ])


def signed_division_loop_program() -> Program:
    """An ENDLESS loop over the wide opcodes on RAW 64-bit registers (AIR mode 4's wide tape; run with max_cycles, halt = CycleLimit): an odd byte b <- b + 2 is stored and loaded
    back with LB — sign-extended to 64 bits when its top bit is set (execute.rs:477-499) — then divided, reduced and multiplied as the reference does it on `as i64` / u64 /
    u128 (quirks Q2, Q3): DIV and REM of a negative by a positive, DIVU of the same bits read as unsigned, MULH of two 64-bit values, REMU and DIV with the wide value as the
    divisor.  Rows whose operands have bits above 40 go through the tape, the others (b < 0x80) through the chunk relation: both routes in one run, 12 rows per iteration."""
    E, O = encode, Opcode
    return Program.from_code([
        addi(6, 0, 0x8000), slli(6, 6, 1), addi(1, 0, 0x95),                                              # base 0x10000, the byte
        # L:
        E(O.SB, rs1=6, rs2=1, imm=0), E(O.LB, 2, 6, imm=0), E(O.ANDI, 3, 1, imm=0x3F), addi(3, 3, 7),      # r2 = sext(byte), a divisor 7..70
        E(O.DIV, 4, 2, 3), E(O.REM, 5, 2, 3), E(O.DIVU, 7, 2, 3), E(O.MULH, 8, 2, 2), E(O.REMU, 9, 3, 2), E(O.DIV, 10, 3, 2),
        addi(1, 1, 2), jal(0, -44),                                                                      # (the byte stays odd: never a zero divisor)
    ])


def memory_ring_program(log2_cells: int = 15) -> Program:
    """An ENDLESS walk over a ring of 2^log2_cells 8-byte cells at 0x100000 (AIR mode 3 at any size: run with max_cycles, halt = CycleLimit): per iteration (16 rows) the
    offset advances by 8 and is wrapped with ANDI, the cell gets the XOR of its old LD value with a counter (SD), is read back as a word, an unsigned halfword and a signed byte
    (LW / LHU / LB), and the running sum is ORed into a flag register — 5 memory accesses and 3 bitwise opcodes in 16 rows, every cell re-visited after 2^log2_cells iterations."""
    assert 3 <= log2_cells <= 13 + 3, "the ring mask must fit a positive 17-bit immediate"
    mask = (8 << log2_cells) - 8
    assert mask <= 65535
    return Program.from_code([
        addi(6, 0, 0x8000), slli(6, 6, 5), addi(1, 0, 0), addi(5, 0, 0), addi(4, 0, 0), addi(12, 0, 0),      # base 0x100000, counter, offset, sum, flags
        # L:
        add(7, 6, 5),                                                             # the cell's address
        encode(Opcode.LD, 2, 7, imm=0), encode(Opcode.XOR, 2, 2, 1), encode(Opcode.SD, rs1=7, rs2=2, imm=0),
        lw(8, 7, 0), encode(Opcode.LHU, 9, 7, imm=2), encode(Opcode.LB, 10, 7, imm=1),
        add(4, 4, 8), add(4, 4, 9), add(4, 4, 10), encode(Opcode.OR, 12, 12, 4),
        addi(5, 5, 8), encode(Opcode.ANDI, 5, 5, imm=mask), addi(1, 1, 1), addi(13, 1, 0),
        jal(0, -60),
    ])


def sha256_chain_program(seed: bytes = bytes(range(32))) -> Program:
    """SHA-256 hash-chain loop of SURVEY.md §8(d) config 5 (pattern of zkir-runtime/tests/crypto_edge_cases.rs:405-427).
    The 32-byte seed is copied from the data section to 0x10000; then forever: sha256(in, 32, out); swap(in, out).
    Every ECALL row carries 32 byte-reads + 8 word-writes; run with max_cycles = 2^k (halt = CycleLimit)."""
    assert len(seed) == 32
    code = [
        addi(5, 0, 0),                                   # [0] r5 = data base (patched below; data sits right after code, vm.rs:164-170)
        addi(6, 0, 0x8000), slli(6, 6, 1),               # r6 = 0x10000
        addi(7, 5, 32),                                  # r7 = end of seed
        lw(8, 5, 0), sw(6, 8, 0), addi(5, 5, 4), addi(6, 6, 4), bne(5, 7, -16),   # [4..8] copy loop
        addi(11, 0, 0x8000), slli(11, 11, 1),            # in  = 0x10000
        addi(13, 11, 32),                                # out = 0x10020
        addi(12, 0, 32),                                 # len = 32
        addi(10, 0, 3), ecall(),                         # [13] L: SYS_SHA256
        addi(9, 11, 0), addi(11, 13, 0), addi(13, 9, 0), # swap in/out
        jal(0, -20),                                     # [18] -> L
    ]
    code[0] = addi(5, 0, CODE_BASE + 4 * len(code))
    return Program.from_code(code, data=seed)
