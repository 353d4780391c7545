"""Pin the CPU oracle against the reference's own known-answer tests (tests/golden/reference_kats.json,
transcribed by tests/golden/make_kats.py with file:line citations), and check the product's host
interpreter against the same vectors.  No GPU needed."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import api as oracle
from zkir_amd import runtime as rt
from zkir_amd.spec import Program

import helpers

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
HALT = {"Ebreak": 0, "Exit": 1, "CycleLimit": 2}


def _blob(p):
    return Program.from_code(p["code"]).to_bytes()


@pytest.mark.parametrize("p", KATS["programs"], ids=[p["name"] for p in KATS["programs"]])
@pytest.mark.parametrize("impl", ["oracle", "oracle_faithful", "product_host"])
def test_reference_program_kats(p, impl):
    cfg = dict(p.get("config", {}))
    blob, inputs = _blob(p), p.get("inputs", [])
    for idx, word in p.get("code_words", {}).items():
        assert p["code"][int(idx)] == word
    if "error" in p:
        if impl == "product_host":
            with pytest.raises(rt.RuntimeError) as e:
                rt.interpret(blob, inputs, rt.VMConfig(**cfg))
            assert e.value.code == p["error"]
        else:
            with pytest.raises(oracle.OracleError) as e:
                oracle.run(blob, inputs, faithful=impl == "oracle_faithful", **cfg)
            assert e.value.code == p["error"]
        return
    if impl == "product_host":
        log = rt.interpret(blob, inputs, rt.VMConfig(**cfg))
        cycles, outputs, halt = log.cycles, list(log.outputs), (log.halt_reason.kind, log.halt_reason.code)
        n_rows = log.n_rows
        rows_cycle = np.arange(n_rows) + log.cycle_base
        memops = helpers.memops_from_log(log)
        offs = np.searchsorted(log.mem_events["row"], np.arange(n_rows + 1)) if n_rows else np.zeros(1, dtype=int)
        n_rc_w = len(log.rc_offsets) - 1
    else:
        r = oracle.run(blob, inputs, faithful=impl == "oracle_faithful", **cfg)
        cycles, outputs, halt = r.cycles, list(r.outputs), (r.halt_kind, r.halt_code if r.halt_kind == 1 else 0)
        n_rows, rows_cycle, memops, offs = len(r.rows), r.rows["cycle"], r.memops, r.row_memop_offsets
        n_rc_w = len(r.rc_offsets) - 1
    if "outputs" in p:
        assert outputs == p["outputs"]
    if "cycles" in p:
        assert cycles == p["cycles"]
    if "halt" in p:
        assert halt == (HALT[p["halt"][0]], p["halt"][1] if len(p["halt"]) > 1 else 0)
    if "n_rows" in p:
        assert n_rows == p["n_rows"]
    if p.get("row_cycle_is_index"):
        assert list(rows_cycle) == list(range(n_rows))
    if "n_memops" in p:
        assert len(memops) == p["n_memops"]
    for row, kinds in p.get("row_memops", {}).items():
        ops = memops[int(offs[int(row)]):int(offs[int(row) + 1])]
        assert ["W" if o["is_write"] else "R" for o in ops] == kinds
    if p.get("memop_ts_is_row"):
        for i in range(n_rows):
            assert (memops[int(offs[i]):int(offs[i + 1])]["timestamp"] == i).all()
    if "rc_witnesses" in p:
        assert n_rc_w == p["rc_witnesses"]
    if "rc_witnesses_min" in p:
        assert n_rc_w >= p["rc_witnesses_min"]


def test_sha256_kats_and_hashlib():
    for k in KATS["sha256"]:
        w = oracle.sha256(bytes.fromhex(k["msg_hex"]))
        if "words" in k:
            assert list(w) == k["words"]
        else:
            assert list(w[:len(k["words_prefix"])]) == k["words_prefix"]
    rng = np.random.default_rng(1)
    for n in [0, 1, 55, 56, 63, 64, 65, 119, 120, 1000]:
        m = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.sha256(m).astype(">u4").tobytes() == hashlib.sha256(m).digest()


def test_keccak_blake3_kats():
    for k in KATS["keccak256"]:
        assert oracle.keccak256(bytes.fromhex(k["msg_hex"])).hex() == k["digest_hex"]
    for k in KATS["blake3"]:
        assert oracle.blake3(bytes.fromhex(k["msg_hex"])).hex() == k["digest_hex"]
    # Keccak-256 is NOT NIST SHA3-256 (different domain byte)
    assert oracle.keccak256(b"") != hashlib.sha3_256(b"").digest()


def test_sha256_witness_kats():
    for k in KATS["sha256_witness"]:
        w = oracle.sha256_witness(bytes.fromhex(k["msg_hex"]), k["timestamp"])
        assert list(w["final_state"]) == k["final_state"]
        assert w["round_states"].shape == (k["n_rounds"], 8)
        assert list(w["message_schedule"][:16]) == list(w["message_block"])        # crypto.rs:772-775
        if "initial_state" in k:
            assert list(w["initial_state"]) == k["initial_state"]
        if "message_block_prefix" in k:
            assert list(w["message_block"][:2]) == k["message_block_prefix"]
        assert not np.array_equal(w["round_states"][0], w["round_states"][63])
    with pytest.raises(oracle.OracleError):
        oracle.sha256_witness(bytes(KATS["sha256_witness_too_long"]["len"]))
    # final state == library digest for every single-block length (crypto.rs:784-834)
    for n in range(56):
        m = bytes((7 * i + n) & 0xFF for i in range(n))
        assert oracle.sha256_witness(m)["final_state"].astype(">u4").tobytes() == hashlib.sha256(m).digest()


def test_decode_kats():
    for k in KATS["decode"]:
        d = oracle.decode(k["word"])
        for f in ("op", "rd", "rs1", "rs2", "imm", "shamt"):
            if f in k:
                # S/B-type: rs1 in bits 10:7, rs2 in bits 14:11
                assert d[f] == k[f], (k, d)
    assert oracle.decode(0x7F) is None and oracle.decode(0x09) is None


def test_value40_m31_kats():
    ops = {"add": 0, "sub": 1, "mul": 2, "shl": 3, "srl": 4, "sra": 5, "slt": 6, "ult": 7}
    for op, a, b, out in KATS["value40"]:
        assert oracle.value40(ops[op], a, b) == out
    assert oracle.value40(0, (1 << 40) - 1, 1) == 0                      # wraps mod 2^40 (value.rs:620-623)
    assert oracle.value40(3, 1, 40) == 0 and oracle.value40(4, 1 << 39, 40) == 0
    assert oracle.value40(5, 1 << 39, 3) == 0xF000000000                # SRA sign bit is bit 39 (Q5)
    assert oracle.value40(6, 1 << 39, 0) == 1                           # 2^39 is negative
    L = oracle.lib()
    for k in KATS["mersenne31"]:
        if k["op"] == "add": assert L.zo_m31_add(k["a"], k["b"]) == k["out"]
        if k["op"] == "sub": assert L.zo_m31_sub(k["a"], k["b"]) == k["out"]
        if k["op"] == "mul_inv": assert L.zo_m31_mul(k["a"], L.zo_m31_inv(k["a"])) == k["out"]
        if k["op"] == "pow": assert L.zo_m31_pow(k["a"], k["b"]) == k["out"]
    for op, a, b, out in KATS["mersenne31_more"]:
        got = {"new": lambda: L.zo_m31_new(a), "add": lambda: L.zo_m31_add(a, b), "sub": lambda: L.zo_m31_sub(a, b),
               "mul": lambda: L.zo_m31_mul(a, b), "neg": lambda: L.zo_m31_neg(a), "pow": lambda: L.zo_m31_pow(a, b),
               "mul_inv": lambda: L.zo_m31_mul(a, L.zo_m31_inv(a))}[op]()
        assert got == out, (op, a, b, got, out)


def test_program_blob_kats():
    k = KATS["program_blob"]
    p = Program()
    assert p.to_bytes()[:4].hex() == k["magic_bytes_hex"] and p.header.version == k["version"]
    assert (p.header.limb_bits, p.header.data_limbs, p.header.addr_limbs) == (20, 2, 2)
    q = Program.from_code(k["roundtrip"]["code"], bytes.fromhex(k["roundtrip"]["data_hex"]))
    r = Program.from_bytes(q.to_bytes())
    assert r.code == q.code and r.data == q.data and r.header == q.header


def test_blob_error_message_kats():
    """The text a ZkIrError displays for a malformed blob (error.rs / config.rs literals) — oracle and product loader, word for word."""
    good = Program().to_bytes()
    cases = [(good[:k["offset"]] + bytes.fromhex(k["bytes_hex"]) + good[k["offset"] + len(k["bytes_hex"]) // 2:], k["message"]) for k in KATS["blob_errors"]]
    cases.append((good[:KATS["blob_truncated"]["keep"]], KATS["blob_truncated"]["message"]))
    for blob, msg in cases:
        with pytest.raises(oracle.OracleError) as eo:
            oracle.run(blob)
        with pytest.raises(rt.RuntimeError) as ep:
            rt.interpret(blob)
        assert eo.value.code == ep.value.code == 7 and eo.value.msg == ep.value.message == msg, (eo.value.msg, ep.value.message, msg)


def test_normalize_and_range_chunk_kats():
    """Drive the deferred model / range checker so the KAT limbs actually occur in a run."""
    from zkir_amd import spec
    from zkir_amd.spec import Opcode as O, encode as E
    A = lambda rd, rs1, imm: E(O.ADDI, rd, rs1, imm=imm)  # noqa: E731
    # normalize.rs:331-360: 32768 + (-16) in limb arithmetic -> accumulated [1081328, 1048575] -> 32752, carries [1,1]
    code = [A(1, 0, 32767), A(1, 1, 1), E(O.ANDI, 1, 1, imm=-1), A(2, 1, -16), E(O.ANDI, 3, 2, imm=-1), spec.ebreak()]
    blob = Program.from_code(code).to_bytes()
    r = oracle.run(blob, enable_deferred_model=True, enable_execution_trace=True)
    ev = r.norm_events[-1]
    k = KATS["normalize"][1]
    assert list(ev["accumulated"]) == k["accumulated"] and list(ev["normalized"]) == k["normalized"] and list(ev["carries"]) == k["carries"]
    log = rt.interpret(blob, config=rt.VMConfig(enable_deferred_model=True, enable_execution_trace=True))
    assert np.array_equal(helpers.norm_from_log(log), r.norm_events)
    # deferred_integration_test.rs:270-316: (2^20 - 10) + 100 -> [90, 1], carry 1
    code = [A(1, 0, 0), E(O.ORI, 1, 1, imm=65535), E(O.SLLI, 1, 1, imm=4), E(O.ORI, 1, 1, imm=6), A(2, 0, 100), spec.add(3, 1, 2),
            E(O.ANDI, 4, 3, imm=-1), spec.ebreak()]
    r = oracle.run(Program.from_code(code).to_bytes(), enable_deferred_model=True)
    ev = r.norm_events[-1]
    k = KATS["normalize"][2]
    assert list(ev["accumulated"]) == k["accumulated"] and list(ev["normalized"]) == k["normalized"] and list(ev["carries"]) == k["carries"]
    # range_check.rs:271-290: value with limbs [0x12345, 0xABCDE] -> chunks [0x345, 0x048, 0x0DE, 0x2AF]
    k = KATS["range_check_chunks"][0]
    value = k["limbs"][0] | (k["limbs"][1] << 20)
    import programs
    code = programs.li40(1, value) + programs.li40(2, 1) + [A(3, 0, 1)] + [spec.add(3, 3, 3)] * 41 + [spec.mul(4, 1, 2), E(O.MUL, 5, 4, 3),
                                                                                                       spec.sw(0, 0, 0x2000), spec.ebreak()]
    blob = Program.from_code(code).to_bytes()
    r = oracle.run(blob, enable_range_checking=True)
    hit = [c for c in r.rc_checks if c["value"] == value]
    assert hit and list(hit[0]["chunks"]) == k["chunks"]
    log = rt.interpret(blob, config=rt.VMConfig(enable_range_checking=True))
    assert np.array_equal(helpers.rc_from_log(log), r.rc_checks) and np.array_equal(log.rc_offsets, r.rc_offsets)
