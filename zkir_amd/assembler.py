"""ZKIR v3.4 assembler (text -> Program), host-only mirror of zkir-assembler.

Follows zkir-assembler/src/{lexer.rs:8-52, parser.rs:11-54, assembler.rs:43-553} including its quirks:
  * `#` comments only; one instruction (optionally after `label:`) per line; mnemonics are case-insensitive;
  * registers r0..r15 or the assembler's ABI names (zero ra sp gp tp fp s0 s1 t0 t1 t2 a0..a4 -> R0..R15, parser.rs:15-45;
    `a5` lexes as a register but is rejected, lexer.rs:16);
  * labels are collected (duplicates / bad names rejected) but NEVER resolved in operands: branch / jump operands must be
    numbers (second_pass ignores `_labels`, assembler.rs:198-209);
  * `.config limb_bits|data_limbs|addr_limbs N` (validated after every directive); every other directive is ignored;
  * numbers: decimal (optionally negative), 0x.., 0b..; converted with `as i32` / `as u8` (wrapping) and then silently masked by the
    encoder to 17 / 21 bits (encoder.rs:117,149);
  * loads `rd, off(rs1)`, stores `rs2, off(rs1)`, branches `rs1, rs2, off`, `jal rd, off`, `jalr rd, rs1, off`.
The result is a `zkir_amd.spec.Program` whose header carries the config and code_size (assembler.rs:50-54).
"""
from __future__ import annotations

import re
from typing import List, Tuple

from . import spec
from .spec import Opcode as O

_REGS = {"zero": 0, "ra": 1, "sp": 2, "gp": 3, "tp": 4, "fp": 5, "s0": 6, "s1": 7, "t0": 8, "t1": 9, "t2": 10,
         "a0": 11, "a1": 12, "a2": 13, "a3": 14, "a4": 15}
_REGS.update({f"r{i}": i for i in range(16)})

# token classes in logos priority: longest match wins; on equal length the more specific pattern (register, number) beats identifier
_TOKEN_RE = re.compile(r"""
    (?P<ws>[ \t]+) |
    (?P<hex>0x[0-9a-fA-F]+) |
    (?P<bin>0b[01]+) |
    (?P<num>-?[0-9]+) |
    (?P<dir>\.[a-zA-Z_][a-zA-Z0-9_]*) |
    (?P<ident>[a-zA-Z_][a-zA-Z0-9_]*) |
    (?P<comma>,) | (?P<colon>:) | (?P<lp>\() | (?P<rp>\))
""", re.X)
_REG_RE = re.compile(r"^(r([0-9]|1[0-5])|zero|ra|sp|gp|tp|fp|s[01]|t[0-2]|a[0-5])$")


class AssemblerError(Exception):
    def __init__(self, line: int, message: str):
        super().__init__(f"line {line}: {message}")
        self.line, self.message = line, message


def _tokenize(text: str, line: int) -> List[Tuple[str, object]]:
    out, pos = [], 0
    while pos < len(text):
        m = _TOKEN_RE.match(text, pos)
        if not m:
            raise AssemblerError(line, f"Invalid token at position {pos}")
        pos = m.end()
        kind = m.lastgroup
        s = m.group(kind)
        if kind == "ws":
            continue
        if kind == "ident":
            out.append(("reg", s) if _REG_RE.match(s) else ("ident", s))
        elif kind == "hex":
            out.append(("num", int(s[2:], 16)))
        elif kind == "bin":
            out.append(("num", int(s[2:], 2)))
        elif kind == "num":
            out.append(("num", int(s)))
        elif kind == "dir":
            out.append(("dir", s[1:]))
        else:
            out.append((kind, s))
    return out


def _i32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def _reg(tok, line):
    if tok[0] != "reg":
        raise AssemblerError(line, f"Expected register, got {tok!r}")
    name = tok[1].lower()
    if name not in _REGS:
        raise AssemblerError(line, f"Invalid register: {name}")
    return _REGS[name]


def parse_register(name: str) -> int:
    """zkir_assembler::parse_register (parser.rs:22-49): r0..r15 and the ABI aliases, case-insensitive -> register index."""
    n = name.lower()
    if n not in _REGS:
        raise AssemblerError(0, f"Invalid register: {name}")
    return _REGS[n]


def _num(tok, line):
    if tok[0] != "num":
        raise AssemblerError(line, f"Expected number, got {tok!r}")
    return tok[1]


def _expect(tok, kind, what, line):
    if tok[0] != kind:
        raise AssemblerError(line, f"Expected {what}, got {tok!r}")


_R = {"add": O.ADD, "sub": O.SUB, "mul": O.MUL, "mulh": O.MULH, "div": O.DIV, "divu": O.DIVU, "rem": O.REM, "remu": O.REMU, "and": O.AND, "or": O.OR,
      "xor": O.XOR, "sll": O.SLL, "srl": O.SRL, "sra": O.SRA, "slt": O.SLT, "sltu": O.SLTU, "sge": O.SGE, "sgeu": O.SGEU, "seq": O.SEQ, "sne": O.SNE,
      "cmov": O.CMOV, "cmovz": O.CMOVZ, "cmovnz": O.CMOVNZ}
_I = {"addi": O.ADDI, "xori": O.XORI, "ori": O.ORI, "andi": O.ANDI}
_SH = {"slli": O.SLLI, "srli": O.SRLI, "srai": O.SRAI}
_LD = {"lw": O.LW, "lh": O.LH, "lhu": O.LHU, "lb": O.LB, "lbu": O.LBU, "ld": O.LD}
_ST = {"sw": O.SW, "sh": O.SH, "sb": O.SB, "sd": O.SD}
_BR = {"beq": O.BEQ, "bne": O.BNE, "blt": O.BLT, "bge": O.BGE, "bltu": O.BLTU, "bgeu": O.BGEU}


def _parse_instruction(toks, line) -> int:
    if toks[0][0] not in ("ident",):
        raise AssemblerError(line, f"Expected instruction mnemonic, got {toks[0]!r}")
    mn, ops = toks[0][1].lower(), toks[1:]

    def three(kinds, msg):
        if len(ops) != 5:
            raise AssemblerError(line, msg)
        _expect(ops[1], "comma", "comma", line); _expect(ops[3], "comma", "comma", line)
        return ops[0], ops[2], ops[4]

    if mn in ("ecall", "ebreak"):
        if ops:
            raise AssemblerError(line, "Instruction takes no operands")
        return spec.ecall() if mn == "ecall" else spec.ebreak()
    if mn in _R:
        a, b, c = three(None, "R-type requires 3 operands: rd, rs1, rs2")
        return spec.encode(_R[mn], _reg(a, line), _reg(b, line), _reg(c, line))
    if mn in _I:
        a, b, c = three(None, "I-type requires 3 operands: rd, rs1, imm")
        return spec.encode(_I[mn], _reg(a, line), _reg(b, line), imm=_i32(_num(c, line)))
    if mn in _SH:
        a, b, c = three(None, "Shift requires 3 operands: rd, rs1, shamt")
        return spec.encode(_SH[mn], _reg(a, line), _reg(b, line), imm=_num(c, line) & 0xFF)           # `as u8`
    if mn in _LD or mn in _ST:
        if len(ops) != 6:
            raise AssemblerError(line, "Load requires format: rd, offset(rs1)" if mn in _LD else "Store requires format: rs2, offset(rs1)")
        r0 = _reg(ops[0], line); _expect(ops[1], "comma", "comma", line)
        off = _i32(_num(ops[2], line)); _expect(ops[3], "lp", "'('", line)
        base = _reg(ops[4], line); _expect(ops[5], "rp", "')'", line)
        if mn in _LD:
            return spec.encode(_LD[mn], r0, base, imm=off)
        return spec.encode(_ST[mn], rs1=base, rs2=r0, imm=off)
    if mn in _BR:
        a, b, c = three(None, "Branch requires 3 operands: rs1, rs2, offset")
        return spec.encode(_BR[mn], rs1=_reg(a, line), rs2=_reg(b, line), imm=_i32(_num(c, line)))
    if mn == "jal":
        if len(ops) != 3:
            raise AssemblerError(line, "JAL requires 2 operands: rd, offset")
        _expect(ops[1], "comma", "comma", line)
        return spec.jal(_reg(ops[0], line), _i32(_num(ops[2], line)))
    if mn == "jalr":
        a, b, c = three(None, "JALR requires 3 operands: rd, rs1, offset")
        return spec.encode(O.JALR, _reg(a, line), _reg(b, line), imm=_i32(_num(c, line)))
    raise AssemblerError(line, f"Invalid instruction: {mn}")


def assemble(source: str) -> spec.Program:
    """assemble(source) of zkir-assembler/src/assembler.rs:43-57."""
    code: List[int] = []
    labels = {}
    cfg = spec.Config()
    pc = spec.CODE_BASE
    for ln, raw in enumerate(source.split("\n"), start=1):
        text = raw.strip()
        if not text or text.startswith("#"):
            continue
        if "#" in text:
            text = text[:text.index("#")].strip()
            if not text:
                continue
        toks = _tokenize(text, ln)
        if not toks:
            continue
        if len(toks) >= 2 and toks[0][0] in ("ident", "reg") and toks[1][0] == "colon" and toks[0][0] == "ident":
            name = toks[0][1]
            if name in labels:
                raise AssemblerError(ln, f"Duplicate label: {name}")
            labels[name] = pc
            if len(toks) > 2:
                code.append(_parse_instruction(toks[2:], ln)); pc += 4
            continue
        if toks[0][0] == "dir":
            if toks[0][1] == "config":
                if len(toks) != 3:
                    raise AssemblerError(ln, ".config requires 2 arguments: key value")
                if toks[1][0] != "ident":
                    raise AssemblerError(ln, "Config key must be an identifier")
                key, value = toks[1][1], _num(toks[2], ln) & 0xFF
                if key not in ("limb_bits", "data_limbs", "addr_limbs"):
                    raise AssemblerError(ln, f"Invalid config value: {key}={value}")
                setattr(cfg, key, value)
                try:
                    cfg.validate()
                except ValueError as e:
                    raise AssemblerError(ln, f"Config error: {e}")
            continue                                   # other directives (.text, .data, ...) are ignored
        code.append(_parse_instruction(toks, ln)); pc += 4
    return spec.Program.from_code(code, config=cfg)
