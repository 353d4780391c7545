#!/usr/bin/env python3
"""Writes tests/golden/stark_goldens.json: frozen outputs of the self-defined prover stages (ZKIR-STARK, AIR v6, proof format v10).

The reference has no prover, so nothing external can pin these stages; this file FREEZES them — it pins nothing:
the values are produced by this repository's own oracle, so they guard against DRIFT only: the
commitment roots and a SHA-256 of the full proof words for four small runs, computed by the CPU oracle (oracle/stark_oracle.cpp).
tests/test_stark_goldens.py checks the oracle (CPU) and the GPU prover (-m gpu) against it, so no change to the field, Poseidon2
instance, main-trace columns, AIR, transcript, FRI schedule, grinding or serialisation can land without this file changing —
deliberately, in its own commit.

Does NOT import the product package: the programs are encoded here from the reference's bit layout (zkir-spec/src/encoding.rs:23-60,
program header zkir-spec/src/program.rs:170-186), like tests/golden/make_kats.py.

Run: python tests/golden/make_stark_goldens.py
"""
import hashlib
import json
import os
import struct
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import api as oracle, stark_api as so  # noqa: E402  (test infrastructure only)

ADD, SUB, MUL, ADDI, SLLI, SLTU, SGEU, SLT, SGE, SEQ, SNE, CMOV, CMOVZ, CMOVNZ, LB, LW, SW, BEQ, BNE, BLT, BGE, BLTU, BGEU, JAL, JALR, ECALL = \
    0x00, 0x01, 0x02, 0x08, 0x1B, 0x20, 0x21, 0x22, 0x23, 0x24, 0x25, 0x26, 0x27, 0x28, 0x30, 0x34, 0x3A, 0x40, 0x41, 0x42, 0x43, 0x44, 0x45, 0x48, 0x49, 0x50


def r_(op, rd, rs1, rs2): return op | rd << 7 | rs1 << 11 | rs2 << 15
def i_(op, rd, rs1, imm): return op | rd << 7 | rs1 << 11 | (imm & 0x1FFFF) << 15
def j_(op, rd, off): return op | rd << 7 | (off & 0x1FFFFF) << 11


def blob(code, data=b""):
    hdr = struct.pack("<IIBBBBIIIII", 0x52494B5A, 0x00030004, 20, 2, 2, 0, 0x1000, 4 * len(code), len(data), 0, 1 << 20)
    return hdr + b"".join(struct.pack("<I", w & 0xFFFFFFFF) for w in code) + data


FIB_LOOP = [r_(ADD, 4, 1, 2), i_(ADDI, 1, 2, 0), i_(ADDI, 2, 4, 0), i_(ADDI, 3, 3, -1), i_(BNE, 3, 0, -16)]      # tests/cross_module.rs:145-164
FIB_ENDLESS = blob([i_(ADDI, 1, 0, 0), i_(ADDI, 2, 0, 1), i_(ADDI, 3, 0, 0)] + FIB_LOOP + [j_(JAL, 0, -20)])
FIB30 = blob([i_(ADDI, 1, 0, 0), i_(ADDI, 2, 0, 1), i_(ADDI, 3, 0, 29)] + FIB_LOOP +
             [i_(ADDI, 11, 2, 0), i_(ADDI, 10, 0, 2), ECALL, i_(ADDI, 10, 0, 0), i_(ADDI, 11, 0, 0), ECALL])
_sha = [i_(ADDI, 5, 0, 0), i_(ADDI, 6, 0, 0x8000), i_(SLLI, 6, 6, 1), i_(ADDI, 7, 5, 32),
        i_(LW, 8, 5, 0), i_(SW, 6, 8, 0), i_(ADDI, 5, 5, 4), i_(ADDI, 6, 6, 4), i_(BNE, 5, 7, -16),
        i_(ADDI, 11, 0, 0x8000), i_(SLLI, 11, 11, 1), i_(ADDI, 13, 11, 32), i_(ADDI, 12, 0, 32),
        i_(ADDI, 10, 0, 3), ECALL, i_(ADDI, 9, 11, 0), i_(ADDI, 11, 13, 0), i_(ADDI, 13, 9, 0), j_(JAL, 0, -20)]
_sha[0] = i_(ADDI, 5, 0, 0x1000 + 4 * len(_sha))
SHA_CHAIN = blob(_sha, bytes(range(32)))                                                                           # pattern of crypto_edge_cases.rs:405-427

# every opcode family of AIR v3 besides the fib loop's, every comparison both ways (the product ships it as spec.compare_loop_program)
CMP_LOOP = blob([i_(ADDI, 1, 0, 0), i_(ADDI, 2, 0, 7), i_(ADDI, 3, 0, 100), i_(ADDI, 10, 0, 1), i_(ADDI, 11, 0, 500),
                 r_(SUB, 4, 3, 2), r_(SUB, 5, 2, 3), r_(SLTU, 6, 2, 11), r_(SGEU, 7, 2, 11), r_(SEQ, 8, 6, 10), r_(SNE, 9, 6, 10),
                 r_(ADD, 2, 2, 6), i_(ADDI, 2, 2, 45), i_(BLTU, 2, 3, 8), r_(ADD, 3, 3, 3), i_(BGEU, 2, 11, 8), r_(SUB, 3, 3, 9),
                 i_(BEQ, 6, 10, 8), i_(ADDI, 1, 1, 1), i_(ADDI, 10, 6, 0), i_(BNE, 1, 0, -60), j_(JAL, 0, -64)])

# a subroutine call loop: JAL / JALR (even and odd return sums, a negative immediate), BLT / BGE, MUL / SLLI (the product ships it as spec.call_loop_program)
CALL_LOOP = blob([i_(ADDI, 1, 0, 0), i_(ADDI, 2, 0, 5), i_(ADDI, 6, 0, 20),
                  j_(JAL, 15, 24), i_(ADDI, 1, 1, 1), i_(BLT, 1, 2, 8), i_(ADDI, 2, 2, 7), i_(BGE, 1, 6, -16), j_(JAL, 0, -20),
                  r_(MUL, 3, 1, 2), i_(SLLI, 4, 3, 2), i_(ADDI, 14, 15, 5), r_(SEQ, 5, 4, 0), i_(BEQ, 5, 0, 8), i_(JALR, 13, 15, 0), i_(JALR, 0, 14, -4)])

# the signed comparisons of AIR v5 on a counter walking from -6 to 8, against thresholds of both signs and the extreme values (the product ships it as spec.signed_loop_program)
SIGNED_LOOP = blob([i_(ADDI, 1, 0, -6), i_(ADDI, 2, 0, -2), i_(ADDI, 3, 0, 3), i_(ADDI, 13, 0, 8), i_(ADDI, 15, 0, 1), i_(SLLI, 15, 15, 39), i_(ADDI, 14, 15, -1),
                    r_(SLT, 4, 1, 2), r_(SGE, 5, 1, 2), r_(SLT, 6, 3, 1), r_(SGE, 7, 3, 1), r_(SLTU, 8, 1, 3), r_(SGEU, 9, 1, 3), r_(SLT, 10, 15, 1), r_(SGE, 11, 14, 1),
                    r_(SLT, 12, 14, 15), i_(BLT, 1, 0, 8), i_(ADDI, 4, 4, 16), i_(BGE, 1, 3, 8), i_(ADDI, 5, 5, 16), i_(BLT, 2, 1, 8), i_(ADDI, 6, 6, 16),
                    i_(BGE, 2, 1, 8), i_(ADDI, 7, 7, 16), i_(BLTU, 1, 3, 8), i_(ADDI, 8, 8, 16), i_(BLT, 14, 15, 8), i_(ADDI, 1, 1, 1), i_(BLT, 1, 13, -84),
                    i_(ADDI, 1, 0, -6), i_(BGE, 14, 15, -92)])

# the conditional moves of AIR v6 with a toggling condition, a raw 64-bit source / condition (sign-extended byte load), rd = r0 (the product ships it as spec.cmov_loop_program)
CMOV_LOOP = blob([i_(ADDI, 1, 0, 0), i_(ADDI, 2, 0, 7), i_(ADDI, 13, 0, 1), i_(ADDI, 10, 0, 0), i_(ADDI, 5, 0, 0x80), i_(ADDI, 6, 0, 0x8000), i_(SLLI, 6, 6, 1), i_(SW, 6, 5, 0),
                  i_(LB, 7, 6, 0), i_(ADDI, 9, 0, 1), i_(SLLI, 9, 9, 20),
                  r_(CMOV, 11, 2, 10), r_(CMOVZ, 12, 7, 10), r_(CMOVNZ, 14, 9, 10), r_(CMOV, 0, 2, 13), r_(CMOVZ, 15, 2, 9), r_(CMOVNZ, 15, 7, 7), r_(CMOVZ, 4, 7, 0),
                  i_(ADDI, 11, 0, 0), i_(ADDI, 12, 0, 0), i_(ADDI, 14, 0, 0), i_(ADDI, 15, 0, 0), i_(ADDI, 4, 0, 0), r_(SUB, 10, 13, 10), i_(ADDI, 1, 1, 1), j_(JAL, 0, -56)])

# (modes 2 / 3) an array loop: SD, then LW / LHU / LB windows of the cell just written, summed and WRITTEN (the product ships it as spec.memory_loop_program(40))
SD, LHU = 0x3B, 0x33
def s_(op, rs1, rs2, imm): return op | rs1 << 7 | rs2 << 11 | (imm & 0x1FFFF) << 15
MEM_LOOP = blob([i_(ADDI, 6, 0, 0x8000), i_(SLLI, 6, 6, 1), i_(ADDI, 1, 0, 0), i_(ADDI, 3, 0, 40), i_(ADDI, 4, 0, 0),
                 r_(ADD, 2, 1, 1), r_(ADD, 2, 2, 1), s_(SD, 6, 2, 0), i_(LW, 7, 6, 0), i_(LHU, 8, 6, 2), i_(LB, 9, 6, 0), r_(ADD, 4, 4, 7), r_(ADD, 4, 4, 8), r_(ADD, 4, 4, 9),
                 i_(ADDI, 6, 6, 8), i_(ADDI, 1, 1, 1), i_(ADDI, 3, 3, -1), i_(BNE, 3, 0, -48),
                 i_(ADDI, 11, 4, 0), i_(ADDI, 10, 0, 2), ECALL, i_(ADDI, 10, 0, 0), i_(ADDI, 11, 0, 0), ECALL])
# (mode 4) the five wide-arithmetic opcodes on a 40-bit pseudo-random state, with one SD / LD per iteration (the product ships it as spec.wide_loop_program)
MUL_, MULH, DIVU, REMU, DIV, REM, XOR, SRLI, ORI, LD_ = 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x12, 0x1C, 0x14, 0x35
WIDE_LOOP = blob([i_(ADDI, 1, 0, 12345), i_(ADDI, 2, 0, 0x2545), i_(SLLI, 2, 2, 4), i_(ADDI, 2, 2, 0xF), i_(ADDI, 12, 0, 0), i_(ADDI, 6, 0, 0x8000), i_(SLLI, 6, 6, 1),
                  r_(MUL_, 1, 1, 2), i_(ADDI, 1, 1, 0x4057), i_(SRLI, 3, 1, 17), i_(ORI, 3, 3, 1),
                  r_(MULH, 4, 1, 1), r_(DIVU, 5, 1, 3), r_(REMU, 7, 1, 3), r_(DIV, 8, 4, 3), r_(REM, 9, 4, 3), r_(MULH, 10, 5, 3),
                  r_(XOR, 12, 12, 4), r_(XOR, 12, 12, 5), r_(XOR, 12, 12, 7), r_(XOR, 12, 12, 8), r_(XOR, 12, 12, 9), r_(XOR, 12, 12, 10),
                  s_(SD, 6, 12, 0), i_(LD_, 13, 6, 0), j_(JAL, 0, -72)])
# (mode 4, the wide tape) the wide opcodes on raw 64-bit registers: a sign-extended byte divided, reduced and multiplied (the product ships it as spec.signed_division_loop_program)
SB_, LB_, ANDI_ = 0x38, 0x30, 0x13
TAPE_LOOP = blob([i_(ADDI, 6, 0, 0x8000), i_(SLLI, 6, 6, 1), i_(ADDI, 1, 0, 0x95),
                  s_(SB_, 6, 1, 0), i_(LB_, 2, 6, 0), i_(ANDI_, 3, 1, 0x3F), i_(ADDI, 3, 3, 7),
                  r_(DIV, 4, 2, 3), r_(REM, 5, 2, 3), r_(DIVU, 7, 2, 3), r_(MULH, 8, 2, 2), r_(REMU, 9, 3, 2), r_(DIV, 10, 3, 2),
                  i_(ADDI, 1, 1, 2), j_(JAL, 0, -44)])
MODE_CASES = [
    dict(name="mode4_signed_division_loop_700", blob=TAPE_LOOP, max_cycles=700, mode=4),
    dict(name="mode4_wide_loop_1000", blob=WIDE_LOOP, max_cycles=1000, mode=4),
    dict(name="mode4_memory_loop_40", blob=MEM_LOOP, max_cycles=1_000_000, mode=4),
    dict(name="mode2_fib30", blob=FIB30, max_cycles=1_000_000, mode=2),
    dict(name="mode3_fib30", blob=FIB30, max_cycles=1_000_000, mode=3),
    dict(name="mode2_memory_loop_40", blob=MEM_LOOP, max_cycles=1_000_000, mode=2),
    dict(name="mode3_memory_loop_40", blob=MEM_LOOP, max_cycles=1_000_000, mode=3),
]

CASES = [
    dict(name="fib_2p10", blob=FIB_ENDLESS, max_cycles=1 << 10, deferred=False),
    dict(name="sha_2p9", blob=SHA_CHAIN, max_cycles=1 << 9, deferred=False),
    dict(name="deferred_fib_2p10", blob=FIB_ENDLESS, max_cycles=1 << 10, deferred=True),
    dict(name="fib30_exit_154_rows", blob=FIB30, max_cycles=1_000_000, deferred=False),
    dict(name="compare_loop_600_rows", blob=CMP_LOOP, max_cycles=600, deferred=False),
    dict(name="call_loop_500_rows", blob=CALL_LOOP, max_cycles=500, deferred=False),
    dict(name="signed_loop_700_rows", blob=SIGNED_LOOP, max_cycles=700, deferred=False),
    dict(name="cmov_loop_400_rows", blob=CMOV_LOOP, max_cycles=400, deferred=False),
]


def golden(case):
    res = oracle.run(case["blob"], max_cycles=case["max_cycles"], enable_execution_trace=True, enable_deferred_model=case["deferred"])
    pub = so.public_inputs(len(res.rows), case["blob"], [], list(res.outputs), (res.halt_kind, res.halt_code), deferred=case["deferred"])
    proof = so.prove(res.rows, pub)
    assert so.verify(proof, pub) == 0
    root = so.commit_trace(res.rows, 1, pub=pub)
    lay = so.proof_layout(proof)
    t0 = lay["trace_root"]
    assert list(root) == list(proof[t0:t0 + 4]) and lay["blob"] == case["blob"]
    return dict(name=case["name"], program_blob_hex=case["blob"].hex(), max_cycles=case["max_cycles"], deferred=case["deferred"],
                n_rows=len(res.rows), outputs=[int(x) for x in res.outputs], halt=[int(res.halt_kind), int(res.halt_code)],
                program_digest=[int(x) for x in pub.prog], io_digest=[int(x) for x in pub.io],
                trace_root=[int(x) for x in proof[t0:t0 + 4]], aux_root=[int(x) for x in proof[t0 + 4:t0 + 8]], quotient_root=[int(x) for x in proof[t0 + 8:t0 + 12]],
                rom_multiplicities=[int(x) for x in proof[lay["rom_mult"]:lay["rc_mult"]]],
                proof_words=int(len(proof)), proof_sha256=hashlib.sha256(proof.astype("<u4").tobytes()).hexdigest())


def golden_mode(case):
    """Modes 2 / 3 / 4 (the I/O argument; + the memory argument; + the wide-arithmetic class, format v12): the whole proof frozen by its SHA-256, the touched cells (modes 3 / 4) by count."""
    res = oracle.run(case["blob"], max_cycles=case["max_cycles"], enable_execution_trace=True)
    pub = so.public_inputs(len(res.rows), case["blob"], [], list(res.outputs), (res.halt_kind, res.halt_code), io_mode=case["mode"] == 2, mem_mode=case["mode"] == 3, wide_mode=case["mode"] == 4)
    proof = so.prove(res.rows, pub)
    assert so.verify(proof, pub) == 0 and int(proof[9]) == case["mode"]
    return dict(name=case["name"], program_blob_hex=case["blob"].hex(), max_cycles=case["max_cycles"], mode=case["mode"], n_rows=len(res.rows), outputs=[int(x) for x in res.outputs],
                halt=[int(res.halt_kind), int(res.halt_code)], committed_width=int(proof[3]), n_cells=int(len(so.mem_cells(res.rows, pub))) if case["mode"] >= 3 else 0,
                proof_words=int(len(proof)), proof_sha256=hashlib.sha256(proof.astype("<u4").tobytes()).hexdigest())


if __name__ == "__main__":
    out = {"_about": "frozen outputs of ZKIR-STARK (AIR v6 + modes 2 / 3, proof format v10; self-defined stages: these values freeze drift, they pin nothing; see make_stark_goldens.py)",
           "proof_version": 10, "main_trace_width": so.W_MAIN, "committed_width": so.W_COMMITTED, "committed_width_deferred": so.W_COMMITTED_DEFERRED, "num_constraints": so.lib().so_num_constraints(),
           "poseidon2_of_0_to_11": [int(x) for x in so.permute(list(range(12)))],
           "cases": [golden(c) for c in CASES], "mode_cases": [golden_mode(c) for c in MODE_CASES]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stark_goldens.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)
    for c in out["cases"] + out["mode_cases"]:
        print(f'  {c["name"]:24s} rows {c["n_rows"]:5d}  root {c.get("trace_root", "-")}  proof {c["proof_words"]} words  sha256 {c["proof_sha256"][:16]}..')
