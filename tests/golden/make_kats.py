#!/usr/bin/env python3
"""Writes tests/golden/reference_kats.json: the known-answer vectors the reference's OWN tests hold for the
execution-trace path, re-typed by hand (the reference is Rust and cannot be run in this image, so these
are transcriptions of literals asserted in its test code — data, not source).  Every entry cites the
reference file:line it was read from (paths relative to /root/reference).

Run: python tests/golden/make_kats.py   (regenerates the JSON next to this file)
"""
import json
import os

# A self-contained encoder, written from the reference's bit layout (zkir-spec/src/encoding.rs:23-60, zkir-assembler/src/
# encoder.rs:100-151): opcode 7 b @0, rd 4 b @7, rs1 4 b @11, rs2 4 b @15 | imm17 @15; S/B-type put rs1 @7, rs2 @11; J: off21 @11.
# Deliberately NOT imported from the product package: the fixtures must not depend on the code they test.  Opcode bytes:
# zkir-spec/src/opcode.rs:154-228.
class O:  # noqa: E742
    ADD, DIVU, DIV, ADDI, LW, SW, BEQ, BNE, JAL, ECALL, EBREAK = 0x00, 0x04, 0x06, 0x08, 0x34, 0x3A, 0x40, 0x41, 0x48, 0x50, 0x51


def E(op, rd=0, rs1=0, rs2=0, imm=0):  # noqa: E743
    if op in (O.ADD, O.DIVU, O.DIV):
        return op | (rd & 0xF) << 7 | (rs1 & 0xF) << 11 | (rs2 & 0xF) << 15
    if op in (O.SW, O.BEQ, O.BNE):
        return op | (rs1 & 0xF) << 7 | (rs2 & 0xF) << 11 | (imm & 0x1FFFF) << 15
    if op == O.JAL:
        return op | (rd & 0xF) << 7 | (imm & 0x1FFFFF) << 11
    if op in (O.ECALL, O.EBREAK):
        return op
    return op | (rd & 0xF) << 7 | (rs1 & 0xF) << 11 | (imm & 0x1FFFF) << 15


class spec:  # the handful of constructors the programs below use, in the reference's `Instruction::X {..}` vocabulary
    add = staticmethod(lambda rd, rs1, rs2: E(O.ADD, rd, rs1, rs2))
    lw = staticmethod(lambda rd, rs1, imm: E(O.LW, rd, rs1, imm=imm))
    sw = staticmethod(lambda rs1, rs2, imm: E(O.SW, rs1=rs1, rs2=rs2, imm=imm))
    beq = staticmethod(lambda rs1, rs2, off: E(O.BEQ, rs1=rs1, rs2=rs2, imm=off))
    bne = staticmethod(lambda rs1, rs2, off: E(O.BNE, rs1=rs1, rs2=rs2, imm=off))
    jal = staticmethod(lambda rd, off: E(O.JAL, rd, imm=off))


assert E(O.ADD, 4, 1, 2) == 0x00010A00 and E(O.BNE, rs1=3, rs2=0, imm=-16) == 0xFFF801C1   # tests/cross_module.rs:140-175 (fib5 words 3 and 7)
A = lambda rd, rs1, imm: E(O.ADDI, rd, rs1, imm=imm)  # noqa: E731
EC, EB = O.ECALL, O.EBREAK
EXIT0 = [A(10, 0, 0), A(11, 0, 0), EC]
WRITE = lambda r: [A(11, r, 0), A(10, 0, 2), EC]  # noqa: E731

programs = [
    # name, cite, code words, inputs, config, expectations
    dict(name="fib5", cite="tests/cross_module.rs:140-175",
         code=[A(1, 0, 0), A(2, 0, 1), A(3, 0, 4), spec.add(4, 1, 2), A(1, 2, 0), A(2, 4, 0), A(3, 3, -1), spec.bne(3, 0, -16)] + WRITE(2) + EXIT0,
         outputs=[5], halt=["Exit", 0], code_words={"3": 0x00010A00, "7": 0xFFF801C1}),
    dict(name="sum_1_to_5", cite="tests/cross_module.rs:406-437",
         code=[A(1, 0, 0), A(2, 0, 1), A(3, 0, 6), spec.add(1, 1, 2), A(2, 2, 1), spec.bne(2, 3, -8)] + WRITE(1) + EXIT0, outputs=[15]),
    dict(name="add_10_20_30", cite="tests/cross_module.rs:60-87",
         code=[A(1, 0, 10), A(2, 0, 20), A(3, 0, 30), spec.add(4, 1, 2), spec.add(4, 4, 3)] + WRITE(4) + EXIT0, outputs=[60]),
    dict(name="branch_skip", cite="tests/cross_module.rs:370-403",
         code=[A(1, 0, 10), A(2, 0, 10), spec.beq(1, 2, 8), A(3, 0, 1), A(4, 0, 2)] + WRITE(3) + EXIT0, outputs=[0]),
    dict(name="sw_lw_roundtrip", cite="tests/cross_module.rs:335-363",
         code=[A(1, 0, 42), A(2, 0, 0x1000), spec.sw(2, 1, 0), spec.lw(3, 2, 0)] + WRITE(3) + EXIT0, outputs=[42]),
    dict(name="echo_123", cite="tests/cross_module.rs:32-57; zkir-runtime/src/vm.rs:489-533",
         code=[A(10, 0, 1), EC, A(11, 10, 0), A(10, 0, 2), EC] + EXIT0, inputs=[123], outputs=[123], halt=["Exit", 0]),
    dict(name="assembled_exit_42", cite="tests/cross_module.rs:15-29; zkir-runtime/src/vm.rs:464-486",
         code=[A(10, 0, 0), A(11, 0, 42), EC], halt=["Exit", 42], cycles=3),
    dict(name="basic_ebreak", cite="zkir-runtime/src/vm.rs:434-461",
         code=[A(1, 0, 10), A(2, 0, 20), spec.add(3, 1, 2), EB], halt=["Ebreak"], cycles=4),
    dict(name="cycle_limit_100", cite="zkir-runtime/src/vm.rs:536-552", code=[spec.jal(0, 0)], config={"max_cycles": 100},
         halt=["CycleLimit"], cycles=100),
    dict(name="memory_operations", cite="zkir-runtime/src/vm.rs:555-591",
         code=[A(1, 0, 0x1000), A(2, 0, 0x42), spec.sw(1, 2, 0), spec.lw(3, 1, 0), EB], halt=["Ebreak"], cycles=5),
    dict(name="branches", cite="zkir-runtime/src/vm.rs:594-632",
         code=[A(1, 0, 10), A(2, 0, 10), spec.beq(1, 2, 8), A(3, 0, 99), EB], halt=["Ebreak"], cycles=4),
    dict(name="thousand_adds", cite="tests/stress_tests.rs:25-56", code=[spec.add(1, 1, 0)] * 1000 + EXIT0, halt=["Exit", 0], cycles=1003),
    dict(name="fifty_nops", cite="tests/stress_tests.rs:162-186", code=[spec.add(0, 0, 0)] * 50 + [EB], config={"max_cycles": 100},
         halt=["Ebreak"], cycles=51),
    dict(name="divu_by_one", cite="tests/stress_tests.rs:437-460",
         code=[A(1, 0, 12345), A(2, 0, 1), E(O.DIVU, 3, 1, 2)] + WRITE(3) + EXIT0, outputs=[12345]),
    dict(name="doubling_80", cite="tests/stress_tests.rs:463-494",
         code=[A(1, 0, 10), spec.add(1, 1, 1), spec.add(1, 1, 1), spec.add(1, 1, 1)] + WRITE(1) + EXIT0, outputs=[80]),
    dict(name="trace_4_rows", cite="zkir-runtime/src/vm.rs:906-964; tests/cross_module.rs:444-468",
         code=[A(1, 0, 100), A(2, 0, 200), spec.add(3, 1, 2), EB], config={"enable_execution_trace": True}, halt=["Ebreak"],
         n_rows=4, row_cycle_is_index=True),
    dict(name="trace_with_memory_ops", cite="zkir-runtime/src/vm.rs:996-1070",
         code=[A(1, 0, 0x42), A(3, 0, 0x1000), spec.sw(3, 1, 0), spec.lw(4, 3, 0), EB], config={"enable_execution_trace": True},
         halt=["Ebreak"], n_rows=5, row_memops={"0": [], "2": ["W"], "3": ["R"]}),
    dict(name="trace_timestamps", cite="zkir-runtime/src/vm.rs:1073-1200",
         code=[A(1, 0, 0x100), A(2, 0, 0x1000), spec.sw(2, 1, 0), A(3, 0, 0x200), spec.sw(2, 3, 4), spec.lw(4, 2, 0), spec.lw(5, 2, 4), EB],
         config={"enable_execution_trace": True}, halt=["Ebreak"], n_memops=4,
         row_memops={"0": [], "1": [], "2": ["W"], "3": [], "4": ["W"], "5": ["R"], "6": ["R"], "7": []}, memop_ts_is_row=True),
    dict(name="trace_disabled", cite="zkir-runtime/src/vm.rs:866-904",
         code=[A(1, 0, 0x42), A(3, 0, 0x1000), spec.sw(3, 1, 0), EB], n_rows=0, n_memops=0),
    dict(name="rc_bound_growth", cite="zkir-runtime/src/vm.rs:698-752",
         code=[A(1, 0, (1 << 15) - 1)] + [spec.add(1, 1, 1)] * 30 + [A(2, 0, 0x1000), spec.sw(2, 1, 0), EB],
         config={"enable_range_checking": True}, halt=["Ebreak"], rc_witnesses_min=1),
    dict(name="rc_small_constants", cite="zkir-runtime/src/vm.rs:755-806",
         code=[A(1, 0, 100), A(2, 0, 200), spec.add(3, 1, 2), A(4, 0, 0x2000), spec.sw(4, 3, 0), EB],
         config={"enable_range_checking": True}, halt=["Ebreak"], rc_witnesses=0),
    dict(name="sha256_empty_syscall", cite="zkir-runtime/tests/crypto_edge_cases.rs:434-496",
         code=[A(1, 0, 0x1000), A(2, 0, 0), A(3, 0, 0x2000), A(10, 0, 3), A(11, 1, 0), A(12, 2, 0), A(13, 3, 0), EC] + EXIT0, halt=["Exit", 0]),
    dict(name="poseidon2_syscall_errors", cite="zkir-runtime/tests/syscall_integration.rs:401-422; zkir-runtime/src/crypto.rs:462-466",
         code=[A(10, 0, 4), EC], error=6),
    dict(name="division_by_zero", cite="zkir-runtime/src/execute.rs:849-867", code=[A(1, 0, 100), E(O.DIV, 3, 1, 2)], error=3),
    dict(name="invalid_syscall_999", cite="zkir-runtime/src/syscall.rs:262-277", code=[A(10, 0, 999), EC], error=4),
]

kats = {
    "_about": "Known-answer vectors transcribed from the reference's own tests; see make_kats.py",
    "programs": programs,
    "sha256": [   # big-endian words read back with LE read_u32, crypto.rs:402-459, crypto_edge_cases.rs:36-127
        {"msg_hex": "", "words": [0xe3b0c442, 0x98fc1c14, 0x9afbf4c8, 0x996fb924, 0x27ae41e4, 0x649b934c, 0xa495991b, 0x7852b855], "cite": "zkir-runtime/src/crypto.rs:402-425"},
        {"msg_hex": b"hello".hex(), "words": [0x2cf24dba, 0x5fb0a30e, 0x26e83b2a, 0xc5b9e29e, 0x1b161e5c, 0x1fa7425e, 0x73043362, 0x938b9824], "cite": "zkir-runtime/src/crypto.rs:428-459"},
        {"msg_hex": b"abc".hex(), "words_prefix": [0xba7816bf, 0x8f01cfea], "cite": "zkir-runtime/tests/crypto_edge_cases.rs:100-127"},
        {"msg_hex": b"a".hex(), "words_prefix": [0xca978112], "cite": "zkir-runtime/tests/crypto_edge_cases.rs:48-63"},
    ],
    "keccak256": [
        {"msg_hex": "", "digest_hex": "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470", "cite": "zkir-runtime/src/crypto.rs:469-490"},
        {"msg_hex": b"hello".hex(), "digest_hex": "1c8aff950685c2ed4bc3174f3472287b56d9517b9c948127319a09a7a36deac8", "cite": "zkir-runtime/src/crypto.rs:493-522"},
    ],
    "blake3": [
        {"msg_hex": "", "digest_hex": "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262", "cite": "zkir-runtime/src/crypto.rs:525-546"},
    ],
    "sha256_witness": [   # crypto.rs:641-711
        {"msg_hex": "", "timestamp": 42, "n_rounds": 64, "final_state": [0xe3b0c442, 0x98fc1c14, 0x9afbf4c8, 0x996fb924, 0x27ae41e4, 0x649b934c, 0xa495991b, 0x7852b855],
         "initial_state": [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], "cite": "zkir-runtime/src/crypto.rs:641-666"},
        {"msg_hex": b"hello".hex(), "timestamp": 100, "n_rounds": 64, "final_state": [0x2cf24dba, 0x5fb0a30e, 0x26e83b2a, 0xc5b9e29e, 0x1b161e5c, 0x1fa7425e, 0x73043362, 0x938b9824],
         "message_block_prefix": [0x68656c6c, 0x6f800000], "cite": "zkir-runtime/src/crypto.rs:669-711"},
    ],
    "sha256_witness_too_long": {"len": 60, "cite": "zkir-runtime/src/crypto.rs:616-638"},
    "range_check_chunks": [   # range_check.rs:271-290: limbs [0x12345, 0xABCDE] -> chunks
        {"limbs": [0x12345, 0xABCDE], "chunks": [0x345, 0x048, 0x0DE, 0x2AF], "cite": "zkir-runtime/src/range_check.rs:271-290"},
    ],
    "normalize": [            # accumulated limbs -> normalized, carries
        {"accumulated": [1048676, 5], "normalized": [100, 6], "carries": [1, 0], "cite": "zkir-runtime/src/normalize.rs:280-302"},
        {"accumulated": [1081328, 1048575], "normalized": [32752, 0], "carries": [1, 1], "value": 32752, "cite": "zkir-runtime/src/normalize.rs:331-360"},
        {"accumulated": [1048666, 0], "normalized": [90, 1], "carries": [1, 0], "cite": "zkir-runtime/src/normalization_witness.rs:289-305; zkir-runtime/tests/deferred_integration_test.rs:270-316"},
        {"accumulated": [3000, 0], "normalized": [3000, 0], "carries": [0, 0], "cite": "zkir-runtime/tests/deferred_integration_test.rs:227-267"},
    ],
    "bounds": [               # bound.rs:435-485, state.rs:361-371
        {"op": "constant", "value": 100, "bits": 7, "cite": "zkir-runtime/src/state.rs:361-371"},
        {"op": "constant", "value": 255, "bits": 8, "cite": "zkir-spec/src/bound.rs:472-478"},
        {"op": "constant", "value": 0x1000, "bits": 13, "cite": "zkir-spec/src/bound.rs:472-478"},
    ],
    "decode": [               # decoder.rs:199-403 (field positions) and sign extension :353-363
        {"word": (0x00 | 1 << 7 | 2 << 11 | 3 << 15), "op": 0x00, "rd": 1, "rs1": 2, "rs2": 3, "cite": "zkir-disassembler/src/decoder.rs:199-216"},
        {"word": (0x08 | 1 << 7 | 2 << 11 | 100 << 15), "op": 0x08, "rd": 1, "rs1": 2, "imm": 100, "cite": "zkir-disassembler/src/decoder.rs:219-236"},
        {"word": (0x34 | 1 << 7 | 2 << 11 | 16 << 15), "op": 0x34, "rd": 1, "rs1": 2, "imm": 16, "cite": "zkir-disassembler/src/decoder.rs:259-276"},
        {"word": (0x3A | 2 << 7 | 1 << 11 | 16 << 15), "op": 0x3A, "rs1": 2, "rs2": 1, "imm": 16, "cite": "zkir-disassembler/src/decoder.rs:279-296"},
        {"word": (0x40 | 1 << 7 | 2 << 11 | 8 << 15), "op": 0x40, "rs1": 1, "rs2": 2, "imm": 8, "cite": "zkir-disassembler/src/decoder.rs:299-316"},
        {"word": (0x48 | 1 << 7 | 100 << 11), "op": 0x48, "rd": 1, "imm": 100, "cite": "zkir-disassembler/src/decoder.rs:319-334"},
        {"word": (0x08 | 1 << 7 | 2 << 11 | (0x1FFFF << 15)) & 0xFFFFFFFF, "op": 0x08, "rd": 1, "rs1": 2, "imm": -1, "cite": "zkir-disassembler/src/decoder.rs:366-383"},
        {"word": (0x1B | 1 << 7 | 2 << 11 | 5 << 15), "op": 0x1B, "rd": 1, "rs1": 2, "shamt": 5, "cite": "zkir-disassembler/src/decoder.rs:386-403"},
        {"word": 0x50, "op": 0x50, "cite": "zkir-disassembler/src/decoder.rs:337-342"},
        {"word": 0x51, "op": 0x51, "cite": "zkir-disassembler/src/decoder.rs:345-350"},
        {"word": (0x08 | (0x10000 << 15)) & 0xFFFFFFFF, "op": 0x08, "imm": -65536, "cite": "zkir-disassembler/src/decoder.rs:353-358"},
        {"word": (0x08 | (0x0FFFF << 15)), "op": 0x08, "imm": 65535, "cite": "zkir-disassembler/src/decoder.rs:353-358"},
        {"word": (0x48 | (0x100000 << 11)) & 0xFFFFFFFF, "op": 0x48, "imm": -1048576, "cite": "zkir-disassembler/src/decoder.rs:360-362"},
    ],
    "opcode_bytes": {"Add": 0x00, "Addi": 0x08, "And": 0x10, "Sll": 0x18, "Sltu": 0x20, "Cmov": 0x26, "Lb": 0x30, "Sb": 0x38, "Beq": 0x40,
                     "Jal": 0x48, "Ecall": 0x50, "_cite": "zkir-disassembler/src/decoder.rs:406-419"},
    "mersenne31": [           # field.rs:239-321
        {"op": "add", "a": 0x7FFFFFFE, "b": 5, "out": 4, "cite": "zkir-spec/src/field.rs:239-321 ((p-1)+5=4)"},
        {"op": "sub", "a": 5, "b": 10, "out": 0x7FFFFFFF - 5, "cite": "zkir-spec/src/field.rs (5-10=p-5)"},
        {"op": "mul_inv", "a": 7, "out": 1, "cite": "zkir-spec/src/field.rs (7*inv(7)=1)"},
        {"op": "pow", "a": 123, "b": 0x7FFFFFFE, "out": 1, "cite": "zkir-spec/src/field.rs (123^(p-1)=1)"},
    ],
    "mersenne31_more": [      # field.rs:231-321, more literals: [op, a, b, out]
        ["new", 0x7FFFFFFF, 0, 0], ["new", 0x80000000, 0, 1], ["new", 0xFFFFFFFE, 0, 0], ["add", 100, 200, 300], ["sub", 200, 100, 100],
        ["mul", 100, 200, 20000], ["neg", 1, 0, 0x7FFFFFFE], ["neg", 0, 0, 0], ["pow", 2, 10, 1024], ["pow", 2, 0, 1], ["mul_inv", 12345, 0, 1],
    ],
    "value40": [              # value.rs:811-893: [op, a, b, out]; ops as in oracle zo_value40_op
        ["add", 100, 50, 150], ["sub", 100, 50, 50], ["mul", 100, 50, 5000], ["shl", 0b1100, 2, 0b110000], ["srl", 0b1100, 1, 0b110],
        ["ult", 50, 100, 1], ["ult", 100, 50, 0],
    ],
    "value40_limbs": {"value": 0x12ABCDE, "limbs": [0xABCDE, 0x12], "cite": "zkir-spec/src/value.rs:885-893"},
    "program_blob": {"magic_bytes_hex": "5a4b4952", "version": 0x00030004, "default_limb_bits": 20, "default_data_limbs": 2, "default_addr_limbs": 2,
                     "roundtrip": {"code": [0x12345678, 0xABCDEF01], "data_hex": "01020304"}, "cite": "zkir-spec/src/program.rs:408-457"},
    "edge_immediates": {"min": -65536, "max": 65535, "cite": "tests/cross_module.rs:229-256"},
    # Program::from_bytes on a default program whose header byte(s) at `offset` are replaced: the Display text of the ZkIrError
    # (zkir-spec/src/error.rs:9-29 with ConfigError's Display, config.rs:215-231; checks in the order of program.rs:147-167, config.rs:154-174)
    "blob_errors": [
        {"offset": 0, "bytes_hex": "00000000", "message": "Invalid program magic: expected 0x5A4B4952, got 0x00000000", "cite": "zkir-spec/src/error.rs:13"},
        {"offset": 4, "bytes_hex": "03000300", "message": "Invalid program version: expected 0x00030004, found 0x00030003", "cite": "zkir-spec/src/error.rs:16"},
        {"offset": 8, "bytes_hex": "0f", "message": "Invalid configuration: limb_bits must be in range [16, 30]", "cite": "zkir-spec/src/config.rs:156-158,218-220"},
        {"offset": 8, "bytes_hex": "20", "message": "Invalid configuration: limb_bits must be in range [16, 30]", "cite": "zkir-spec/src/config.rs:156-158"},
        {"offset": 8, "bytes_hex": "15", "message": "Invalid configuration: limb_bits must be even", "cite": "zkir-spec/src/config.rs:159-161,221-223"},
        {"offset": 9, "bytes_hex": "00", "message": "Invalid configuration: data_limbs must be in range [1, 4]", "cite": "zkir-spec/src/config.rs:164-166,224-226"},
        {"offset": 9, "bytes_hex": "05", "message": "Invalid configuration: data_limbs must be in range [1, 4]", "cite": "zkir-spec/src/config.rs:164-166"},
        {"offset": 10, "bytes_hex": "03", "message": "Invalid configuration: addr_limbs must be in range [1, 2]", "cite": "zkir-spec/src/config.rs:169-171,227-229"},
    ],
    "blob_truncated": {"keep": 31, "message": "Invalid header size: expected 32 bytes, found 31 bytes", "cite": "zkir-spec/src/error.rs:19; program.rs:189-194"},
}

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
with open(out, "w") as f:
    json.dump(kats, f, indent=1, sort_keys=True)
print("wrote", out, len(programs), "programs")
