#!/usr/bin/env python3
"""Writes tests/golden/config_proofs.json: the COMPLETE proof (format v10, mode 0) of the fib_endless run halted at 2^k cycles — k = 20 is the run BASELINE's metric
"end-to-end prove ms, 2^20-cycle fib" is quoted on — computed by the CPU ORACLE alone (oracle/zkir_oracle.cpp executes, oracle/stark_oracle.cpp proves: textbook
arithmetic, one thread, ~13 minutes and ~5 GB at 2^20).  Kept per size: the proof's length in words, the SHA-256 of its little-endian words, and 257 evenly spaced
words (so that a mismatch says roughly WHERE).  tests/test_gpu_large.py::test_proof_equals_the_oracles_at_config_size compares the GPU prover's proof with it:
"proof bytes bit-identical" at the size the metric names, where tests/test_gpu_stark.py compares whole proofs up to 2^13 rows.

What this pins: nothing outside this repository — the prover is self-defined (the reference has none): parity stays UNPINNED vs seceq/zkir.

Also (`python tests/golden/make_config_proofs.py mode3 [log2_rows ...]`, default 14 16): the MODE-3 proof (memory argument, bitwise opcodes, shifts, MUL: 264 + 96 columns,
the touched cells carried) of the memory-ring walk halted at 2^k cycles (1024 cells: every cell re-visited 4 / 16 times; 5 memory accesses and 3 bitwise opcodes per 16
rows) — the large-size counterpart of tests/test_gpu_stark.py's small mode-3 programs; the program is encoded HERE from the reference's bit layout (encoding.rs:23-60).

Also (`python tests/golden/make_config_proofs.py mode2 [log2_rows ...]`, default 12 16 20; round 5): the MODE-2 proof (the I/O argument) of a fib run that halts by itself and
WRITES its result — the proof whose output tape says what the run computed.

Does not import the product.  Run: python tests/golden/make_config_proofs.py [log2_rows ...]   (default 16 18 20; 22: 45 minutes, 23: 104 minutes and ~36 GB, one thread)
2^24 rows (BASELINE configs[2]'s own size; round 6) goes through so::prove_lean — the memory-lean, threaded restatement of so::prove that tests/test_stark_oracle.py holds
equal to it word for word in every mode (38 GB instead of ~70; ZKIR_ORACLE_THREADS std::threads, default all cores) — and so does any size when ZKIR_ORACLE_LEAN=1.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from make_config_roots import FIB_ENDLESS, blob, i_, j_, r_  # noqa: E402  (the same program blob, own encoder)
from oracle import api as oracle, stark_api as so  # noqa: E402  (test infrastructure only)

N_SAMPLES = 257

ADD, ADDI, JAL, OR, XOR, ANDI, SLLI, LB, LHU, LW, LD, SD = 0x00, 0x08, 0x48, 0x11, 0x12, 0x13, 0x1B, 0x30, 0x33, 0x34, 0x35, 0x3B


def s_(op, rs1, rs2, imm): return op | rs1 << 7 | rs2 << 11 | (imm & 0x1FFFF) << 15      # S-type: rs1 @7, rs2 @11
def sh_(op, rd, rs1, sh): return op | rd << 7 | rs1 << 11 | (sh & 0xFF) << 15           # shift by immediate: shamt = bits 15..22


def ring(log2_cells):
    """An endless walk over a ring of 2^log2_cells 8-byte cells at 0x100000: LD, XOR with a counter, SD, read back as LW / LHU / LB, sum, OR into a flag register."""
    mask = (8 << log2_cells) - 8
    return blob([i_(ADDI, 6, 0, 0x8000), sh_(SLLI, 6, 6, 5), i_(ADDI, 1, 0, 0), i_(ADDI, 5, 0, 0), i_(ADDI, 4, 0, 0), i_(ADDI, 12, 0, 0),
                 r_(ADD, 7, 6, 5), i_(LD, 2, 7, 0), r_(XOR, 2, 2, 1), s_(SD, 7, 2, 0), i_(LW, 8, 7, 0), i_(LHU, 9, 7, 2), i_(LB, 10, 7, 1),
                 r_(ADD, 4, 4, 8), r_(ADD, 4, 4, 9), r_(ADD, 4, 4, 10), r_(OR, 12, 12, 4), i_(ADDI, 5, 5, 8), i_(ANDI, 5, 5, mask), i_(ADDI, 1, 1, 1), i_(ADDI, 13, 1, 0),
                 j_(JAL, 0, -60)])


RING = ring(10)


def entry(proof, k, t0):
    pos = sample_positions(len(proof))
    return {"rows": 1 << k, "words": int(len(proof)), "sha256": hashlib.sha256(proof.tobytes()).hexdigest(), "samples": [int(proof[p]) for p in pos], "oracle_seconds": round(time.time() - t0, 1)}


def main_mode3(ks):
    path = os.path.join(HERE, "config_proofs.json")
    for k in ks:
        t0 = time.time()
        res = oracle.run(RING, max_cycles=1 << k, enable_execution_trace=True)
        pub = so.public_inputs(len(res.rows), RING, [], list(res.outputs), (res.halt_kind, res.halt_code), mem_mode=True)
        proof = np.ascontiguousarray(so.prove(res.rows, pub), dtype="<u4")
        assert so.verify(proof, pub) == 0 and int(proof[9]) == 3
        e = entry(proof, k, t0)
        print("mode3", k, e["words"], e["sha256"], e["oracle_seconds"], flush=True)
        cur = json.load(open(path))
        cur["ring_program_blob_hex"] = RING.hex()
        cur.setdefault("mode3_ring_proofs", {})[str(k)] = e
        json.dump(cur, open(path, "w"), indent=1)


ADD_, BNE_, ECALL_ = 0x00, 0x41, 0x50


def b_(op, rs1, rs2, off): return op | rs1 << 7 | rs2 << 11 | (off & 0x1FFFF) << 15      # B-type: rs1 @7, rs2 @11, offset @15


def fib_out(cnt):
    """The v3.4 fib of tests/cross_module.rs:145-164 with a loop counter beyond the 17-bit immediate (Q11): r3 = hi, doubled twelve times, + lo; `cnt` iterations, then
    WRITE r2 (syscall 2: the value leaves on the output tape) and EXIT 0.  16 + 5 cnt + 6 rows."""
    hi, lo = cnt >> 12, cnt & 0xFFF
    return blob([i_(ADDI, 1, 0, 0), i_(ADDI, 2, 0, 1), i_(ADDI, 3, 0, hi)] + [r_(ADD_, 3, 3, 3)] * 12 + [i_(ADDI, 3, 3, lo)]
                + [r_(ADD_, 4, 1, 2), i_(ADDI, 1, 2, 0), i_(ADDI, 2, 4, 0), i_(ADDI, 3, 3, -1), b_(BNE_, 3, 0, -16)]
                + [i_(ADDI, 11, 2, 0), i_(ADDI, 10, 0, 2), ECALL_, i_(ADDI, 10, 0, 0), i_(ADDI, 11, 0, 0), ECALL_])


def main_mode2(ks):
    """MODE-2 proofs (mode 0 + the I/O argument: the proof says what the run WROTE): fib_out with the largest iteration count whose run fits 2^k rows — the run halts by
    itself (Exit 0) a few rows short of 2^k, so the padding rows are exercised too; its one output, fib(cnt + 1) mod 2^40, is in the proof's output tape."""
    path = os.path.join(HERE, "config_proofs.json")
    for k in ks:
        cnt = ((1 << k) - 22) // 5
        prog = fib_out(cnt)
        t0 = time.time()
        res = oracle.run(prog, max_cycles=1 << (k + 1), enable_execution_trace=True)
        assert res.halt_kind == 1 and res.halt_code == 0 and len(res.rows) == 22 + 5 * cnt and len(res.outputs) == 1
        pub = so.public_inputs(len(res.rows), prog, [], list(res.outputs), (res.halt_kind, res.halt_code), io_mode=True)
        proof = np.ascontiguousarray(so.prove(res.rows, pub), dtype="<u4")
        assert so.verify(proof, pub) == 0 and int(proof[9]) == 2
        e = entry(proof, k, t0)
        e.update({"rows": len(res.rows), "iterations": cnt, "output": int(res.outputs[0]), "program_blob_hex": prog.hex()})
        print("mode2", k, e["rows"], e["output"], e["words"], e["sha256"], e["oracle_seconds"], flush=True)
        cur = json.load(open(path))
        cur.setdefault("mode2_fib_out_proofs", {})[str(k)] = e
        json.dump(cur, open(path, "w"), indent=1)


def main_params(ks, num_queries=84, pow_bits=16):
    """The fib_endless run at the SECOND parameter set (round 5: zkir_prover_params — 84 FRI queries + 16 grinding bits, conjectured 100 bits of FRI soundness at blow-up 2;
    the capacity-4 sponge still caps collisions at ~62 bits): mode-0 proofs whose header words 4 and 6 say so."""
    path = os.path.join(HERE, "config_proofs.json")
    for k in ks:
        t0 = time.time()
        res = oracle.run(FIB_ENDLESS, max_cycles=1 << k, enable_execution_trace=True)
        pub = so.public_inputs(len(res.rows), FIB_ENDLESS, [], list(res.outputs), (res.halt_kind, res.halt_code), num_queries=num_queries, pow_bits=pow_bits)
        proof = np.ascontiguousarray(so.prove(res.rows, pub), dtype="<u4")
        assert so.verify(proof, pub) == 0 and int(proof[4]) == num_queries and int(proof[6]) == pow_bits
        assert so.verify(proof, so.public_inputs(len(res.rows), FIB_ENDLESS, [], list(res.outputs), (res.halt_kind, res.halt_code))) == 2      # a verifier expecting the defaults refuses it
        e = entry(proof, k, t0)
        e.update({"num_queries": num_queries, "pow_bits": pow_bits})
        print("params", k, e["words"], e["sha256"], e["oracle_seconds"], flush=True)
        cur = json.load(open(path))
        cur.setdefault("fib_84q_16b_proofs", {})[str(k)] = e
        json.dump(cur, open(path, "w"), indent=1)


def sample_positions(n_words: int):
    return [int(i * (n_words - 1) // (N_SAMPLES - 1)) for i in range(N_SAMPLES)]


def main():
    if sys.argv[1:2] == ["mode3"]:
        return main_mode3([int(a) for a in sys.argv[2:]] or [14, 16])
    if sys.argv[1:2] == ["params"]:
        return main_params([int(a) for a in sys.argv[2:]] or [12, 16])
    if sys.argv[1:2] == ["mode2"]:
        return main_mode2([int(a) for a in sys.argv[2:]] or [12, 16, 20])
    ks = [int(a) for a in sys.argv[1:]] or [16, 18, 20]
    path = os.path.join(HERE, "config_proofs.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["_about"] = ("whole proofs (ZKIR-STARK format v10, mode 0) of the fib_endless run halted at 2^k cycles, written by tests/golden/make_config_proofs.py from the CPU "
                     "oracle alone: length, SHA-256 of the little-endian u32 words, 257 evenly spaced words; self-defined prover: parity UNPINNED vs seceq/zkir (it has "
                     "no prover) — the file makes GPU proof == independent CPU proof literal at the size BASELINE's metric names (2^20 cycles)")
    out["program_blob_hex"] = FIB_ENDLESS.hex()
    proofs = out.setdefault("proofs", {})
    for k in ks:
        t0 = time.time()
        res = oracle.run(FIB_ENDLESS, max_cycles=1 << k, enable_execution_trace=True)
        pub = so.public_inputs(len(res.rows), FIB_ENDLESS, [], list(res.outputs), (res.halt_kind, res.halt_code))
        lean = k >= 24 or os.environ.get("ZKIR_ORACLE_LEAN") == "1"
        threads = int(os.environ.get("ZKIR_ORACLE_THREADS", os.cpu_count() or 1)) if lean else 1
        rows = res.rows
        del res
        proof = np.ascontiguousarray(so.prove_lean(rows, pub, threads=threads) if lean else so.prove(rows, pub), dtype="<u4")
        assert so.verify(proof, pub) == 0
        pos = sample_positions(len(proof))
        proofs[str(k)] = {"rows": 1 << k, "words": int(len(proof)), "sha256": hashlib.sha256(proof.tobytes()).hexdigest(),
                          "samples": [int(proof[p]) for p in pos], "oracle_seconds": round(time.time() - t0, 1), "threads": threads, "prover": "so::prove_lean" if lean else "so::prove"}
        print(k, proofs[str(k)]["words"], proofs[str(k)]["sha256"], proofs[str(k)]["oracle_seconds"], flush=True)
        cur = json.load(open(path)) if os.path.exists(path) else {}          # (several sizes may be running side by side: merge, do not overwrite)
        cur.update({"_about": out["_about"], "program_blob_hex": out["program_blob_hex"]})
        cur.setdefault("proofs", {})[str(k)] = proofs[str(k)]
        json.dump(cur, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
