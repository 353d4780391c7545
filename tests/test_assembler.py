"""v3.4 assembler mirror: sources from the reference's own tests assemble to the same words the
Instruction-literal encoder gives, run to the asserted outputs, and malformed input is rejected."""
import pytest

from oracle import api as oracle
from zkir_amd import runtime as rt, spec
from zkir_amd.assembler import AssemblerError, assemble
from zkir_amd.spec import Opcode as O, encode as E

FIB5 = """
    .config limb_bits 20
    .config data_limbs 2

    # Fibonacci sequence - compute f(5)
    addi r1, zero, 0      # f(0) = 0
    addi r2, zero, 1      # f(1) = 1
    addi r3, zero, 4      # counter (4 iterations: f(2), f(3), f(4), f(5))

loop:
    add r4, r1, r2        # f(n) = f(n-1) + f(n-2)
    addi r1, r2, 0        # f(n-2) = f(n-1)
    addi r2, r4, 0        # f(n-1) = f(n)
    addi r3, r3, -1       # counter--
    bne r3, zero, -16     # loop if counter != 0

    # Write result
    addi a0, r2, 0        # a0 = result (R11)
    addi t2, zero, 2      # syscall: write (R10)
    ecall

    # Exit
    addi t2, zero, 0      # syscall: exit (R10)
    addi a0, zero, 0      # exit code (R11)
    ecall
"""   # tests/cross_module.rs:140-175


def test_fib5_source_words_and_output():
    p = assemble(FIB5)
    assert p.code == spec.fib_program(5).code
    assert p.code[3] == 0x00010A00 and p.code[7] == 0xFFF801C1
    assert p.header.code_size == 4 * 14 and (p.header.limb_bits, p.header.data_limbs) == (20, 2)
    assert rt.run(p) == [5]
    assert list(oracle.run(p.to_bytes()).outputs) == [5]


def test_memory_roundtrip_source():               # tests/cross_module.rs:335-363: `sw r1, 0(r2)` stores rs2=r1 at base rs1=r2
    src = """
        addi r1, zero, 42
        addi r2, zero, 0x1000
        sw r1, 0(r2)
        lw r3, 0(r2)
        addi a0, r3, 0
        addi t2, zero, 2
        ecall
        addi t2, zero, 0
        addi a0, zero, 0
        ecall
    """
    p = assemble(src)
    assert p.code[2] == spec.sw(2, 1, 0) and p.code[3] == spec.lw(3, 2, 0)
    assert rt.run(p) == [42]


def test_every_mnemonic_matches_the_literal_encoder():
    cases = {"add r4, r5, r6": E(O.ADD, 4, 5, 6), "sub r1, r2, r3": E(O.SUB, 1, 2, 3), "mulh r1, r2, r3": E(O.MULH, 1, 2, 3),
             "DIVU r1, r2, r3": E(O.DIVU, 1, 2, 3), "remu r7, r8, r9": E(O.REMU, 7, 8, 9), "cmovnz r1, r2, r3": E(O.CMOVNZ, 1, 2, 3),
             "sgeu r1, r2, r3": E(O.SGEU, 1, 2, 3), "xori r1, r2, -1": E(O.XORI, 1, 2, imm=-1), "andi r1, r2, 0xFFFF": E(O.ANDI, 1, 2, imm=0xFFFF),
             "ori t0, t1, 0b101": E(O.ORI, 8, 9, imm=5), "slli r1, r2, 5": E(O.SLLI, 1, 2, imm=5), "srai r1, r2, 39": E(O.SRAI, 1, 2, imm=39),
             "lb r1, -4(sp)": E(O.LB, 1, 2, imm=-4), "lhu r1, 2(gp)": E(O.LHU, 1, 3, imm=2), "ld a4, 8(fp)": E(O.LD, 15, 5, imm=8),
             "sd s0, 16(s1)": E(O.SD, rs1=7, rs2=6, imm=16), "sb r1, 0(r2)": E(O.SB, rs1=2, rs2=1, imm=0),
             "bgeu r1, r2, 8": E(O.BGEU, rs1=1, rs2=2, imm=8), "blt ra, tp, -4": E(O.BLT, rs1=1, rs2=4, imm=-4),
             "jal ra, 100": spec.jal(1, 100), "jal zero, -20": spec.jal(0, -20), "jalr r1, r2, 12": E(O.JALR, 1, 2, imm=12),
             "ecall": spec.ecall(), "ebreak": spec.ebreak()}
    for src, word in cases.items():
        assert assemble(src).code == [word], src


def test_immediate_edges_are_masked_like_the_encoder():      # tests/cross_module.rs:229-256, encoder.rs:117 (Q11)
    assert assemble("addi r1, r0, 65535").code == [E(O.ADDI, 1, 0, imm=65535)]
    assert assemble("addi r1, r0, -65536").code == [E(O.ADDI, 1, 0, imm=-65536)]
    assert assemble("addi r1, r0, 0x10000").code == [E(O.ADDI, 1, 0, imm=-65536)]      # silently wraps to the 17-bit field
    w = assemble("addi r1, r0, 0x10000").code[0]
    assert oracle.decode(w)["imm"] == -65536


def test_labels_comments_directives():
    p = assemble("start: addi r1, r0, 1   # trailing comment\n.text\n.data\n# whole-line comment\nend:\n  ebreak\n")
    assert p.code == [E(O.ADDI, 1, 0, imm=1), spec.ebreak()]
    p = assemble(".config limb_bits 30\n.config data_limbs 3\n.config addr_limbs 1\necall")
    assert (p.header.limb_bits, p.header.data_limbs, p.header.addr_limbs) == (30, 3, 1)


@pytest.mark.parametrize("src", [
    "foo r1, r2, r3",              # unknown mnemonic
    "add r1, r2",                  # missing operand
    "add r1 r2 r3",                # missing commas
    "add r1, r2, r16",             # r16 is not a register token
    "add r1, r2, a5",              # lexes as a register, rejected by parse_register (parser.rs:47)
    "add r1, r2, t3",              # t3 is not in the assembler's alias table
    "addi r1, r2, r3",             # immediate expected
    "lw r1, r2",                   # load needs off(base)
    "bne r3, zero, loop",          # labels are never resolved in operands (assembler.rs:198-209)
    "x: ebreak\nx: ebreak",        # duplicate label
    ".config limb_bits 15\necall",  # invalid configuration (config.rs:154-174)
    ".config limb_bits 21\necall",
    ".config colour 3\necall",
    ".config limb_bits\necall",
    "ecall r1",
    "addi r1, r0, 5 ; semicolon comments are v2.2 syntax",
])
def test_malformed_input_is_rejected(src):
    with pytest.raises(AssemblerError):
        assemble(src)
