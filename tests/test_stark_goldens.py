"""Frozen goldens of the self-defined prover stages (tests/golden/stark_goldens.json, written by make_stark_goldens.py without the
product package): the CPU oracle and — on the GPU box — the HIP prover must reproduce the commitment roots and the SHA-256 of every
proof word.  A change to the field, the Poseidon2 instance, the main-trace columns, the AIR, the transcript, FRI or the proof layout
shows up here first, and can only land together with a deliberate regeneration of the JSON."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import api as oracle, stark_api as so
from zkir_amd import runtime as rt, spec

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stark_goldens.json")))
CASES = {c["name"]: c for c in G["cases"]}


def test_goldens_describe_the_shipped_programs():
    """The fixture's blobs are the workloads the product ships (encoded there independently, from the reference's bit layout)."""
    assert bytes.fromhex(CASES["fib_2p10"]["program_blob_hex"]) == spec.fib_endless_program().to_bytes()
    assert bytes.fromhex(CASES["sha_2p9"]["program_blob_hex"]) == spec.sha256_chain_program().to_bytes()
    assert bytes.fromhex(CASES["fib30_exit_154_rows"]["program_blob_hex"]) == spec.fib_program(30).to_bytes()
    assert bytes.fromhex(CASES["compare_loop_600_rows"]["program_blob_hex"]) == spec.compare_loop_program().to_bytes()
    assert G["committed_width"] == so.W_COMMITTED == so.committed_width(False) == 152 and G["committed_width_deferred"] == so.committed_width(True) == 168
    assert bytes.fromhex(CASES["call_loop_500_rows"]["program_blob_hex"]) == spec.call_loop_program().to_bytes()
    assert bytes.fromhex(CASES["signed_loop_700_rows"]["program_blob_hex"]) == spec.signed_loop_program().to_bytes()
    assert bytes.fromhex(CASES["cmov_loop_400_rows"]["program_blob_hex"]) == spec.cmov_loop_program().to_bytes()
    assert G["proof_version"] == 10 and G["main_trace_width"] == so.W_MAIN == 172 and G["num_constraints"] == so.lib().so_num_constraints()
    assert [int(x) for x in so.permute(list(range(12)))] == G["poseidon2_of_0_to_11"]
    st = np.arange(12, dtype=np.uint32)
    rt.lib().zkir_poseidon2_permute(st.ctypes.data)                                  # the product's host permutation
    assert [int(x) for x in st] == G["poseidon2_of_0_to_11"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_goldens(name):
    c = CASES[name]
    blob = bytes.fromhex(c["program_blob_hex"])
    res = oracle.run(blob, max_cycles=c["max_cycles"], enable_execution_trace=True, enable_deferred_model=c["deferred"])
    assert len(res.rows) == c["n_rows"] and [int(x) for x in res.outputs] == c["outputs"] and [res.halt_kind, res.halt_code] == c["halt"]
    pub = so.public_inputs(len(res.rows), blob, [], c["outputs"], tuple(c["halt"]), deferred=c["deferred"])
    assert [int(x) for x in pub.prog] == c["program_digest"] and [int(x) for x in pub.io] == c["io_digest"]
    proof = so.prove(res.rows, pub)
    t0 = so.proof_layout(proof)["trace_root"]
    assert int(proof[1]) == 10 and [int(x) for x in proof[t0:t0 + 4]] == c["trace_root"] and [int(x) for x in proof[t0 + 4:t0 + 8]] == c["aux_root"]
    assert [int(x) for x in proof[t0 + 8:t0 + 12]] == c["quotient_root"]
    assert len(proof) == c["proof_words"] and hashlib.sha256(proof.astype("<u4").tobytes()).hexdigest() == c["proof_sha256"]
    assert rt.verify(proof) == 0                                                      # the product's verifier accepts the frozen proofs


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_prover_reproduces_goldens(name):
    from zkir_amd import stark
    c = CASES[name]
    blob = bytes.fromhex(c["program_blob_hex"])
    cfg = rt.VMConfig(max_cycles=c["max_cycles"], enable_execution_trace=True, enable_deferred_model=c["deferred"])
    res = rt.VM(blob, [], cfg).run()
    assert res.cycles == c["n_rows"] and list(res.outputs) == c["outputs"]
    pub = res.public_inputs()
    assert list(pub.program_digest) == c["program_digest"] and list(pub.io_digest) == c["io_digest"]
    ctx = stark.StarkContext(stark.padded_log_n(res.cycles))
    proof = stark.prove(ctx, res.execution_trace.columns, pub)
    t0 = so.proof_layout(proof)["trace_root"]
    assert [int(x) for x in proof[t0:t0 + 4]] == c["trace_root"] and [int(x) for x in proof[t0 + 4:t0 + 8]] == c["aux_root"] and [int(x) for x in proof[t0 + 8:t0 + 12]] == c["quotient_root"]
    assert len(proof) == c["proof_words"] and hashlib.sha256(proof.astype("<u4").tobytes()).hexdigest() == c["proof_sha256"]
    assert rt.verify(proof, pub) == 0
    ctx.close(); res.close()


# ---- modes 2 / 3 (round 4: the I/O argument; + the memory argument): the whole proof frozen by its SHA-256 ----------------------------------------------------
MODE_CASES = {c["name"]: c for c in G["mode_cases"]}


def test_mode_goldens_describe_the_shipped_programs():
    assert bytes.fromhex(MODE_CASES["mode3_memory_loop_40"]["program_blob_hex"]) == spec.memory_loop_program(40).to_bytes()
    assert bytes.fromhex(MODE_CASES["mode3_fib30"]["program_blob_hex"]) == spec.fib_program(30).to_bytes()
    assert [MODE_CASES[n]["committed_width"] for n in ("mode2_fib30", "mode3_fib30")] == [so.committed_width(2), so.committed_width(3)] == [160, 264]
    assert bytes.fromhex(MODE_CASES["mode4_wide_loop_1000"]["program_blob_hex"]) == spec.wide_loop_program().to_bytes() and MODE_CASES["mode4_wide_loop_1000"]["committed_width"] == so.committed_width(4) == 288
    assert bytes.fromhex(MODE_CASES["mode4_signed_division_loop_700"]["program_blob_hex"]) == spec.signed_division_loop_program().to_bytes()      # (the wide tape: raw 64-bit operands)


@pytest.mark.parametrize("name", sorted(MODE_CASES))
def test_oracle_reproduces_mode_goldens(name):
    c = MODE_CASES[name]
    blob = bytes.fromhex(c["program_blob_hex"])
    res = oracle.run(blob, max_cycles=c["max_cycles"], enable_execution_trace=True)
    assert len(res.rows) == c["n_rows"] and [int(x) for x in res.outputs] == c["outputs"] and [res.halt_kind, res.halt_code] == c["halt"]
    pub = so.public_inputs(len(res.rows), blob, [], c["outputs"], tuple(c["halt"]), io_mode=c["mode"] == 2, mem_mode=c["mode"] == 3, wide_mode=c["mode"] == 4)
    proof = so.prove(res.rows, pub)
    assert int(proof[9]) == c["mode"] and int(proof[3]) == c["committed_width"]
    assert len(proof) == c["proof_words"] and hashlib.sha256(proof.astype("<u4").tobytes()).hexdigest() == c["proof_sha256"]
    assert rt.verify(proof) == 0 and so.verify(proof, pub) == 0
    if c["mode"] >= 3:
        assert len(so.mem_cells(res.rows, pub)) == c["n_cells"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODE_CASES))
def test_gpu_prover_reproduces_mode_goldens(name):
    from zkir_amd import stark
    c = MODE_CASES[name]
    blob = bytes.fromhex(c["program_blob_hex"])
    res = rt.VM(blob, [], rt.VMConfig(max_cycles=c["max_cycles"], enable_execution_trace=True)).run()
    assert res.cycles == c["n_rows"] and list(res.outputs) == c["outputs"]
    pub = rt.public_inputs(res._log, blob, [], io_mode=c["mode"] == 2, mem_mode=c["mode"] == 3, wide_mode=c["mode"] == 4)
    ctx = stark.StarkContext(stark.padded_log_n(res.cycles))
    proof = stark.prove(ctx, res.execution_trace.columns, pub)
    assert len(proof) == c["proof_words"] and hashlib.sha256(proof.astype("<u4").tobytes()).hexdigest() == c["proof_sha256"]
    assert rt.verify(proof, pub) == 0
    ctx.close(); res.close()


def test_config_proof_fixture_is_the_oracles_proof():
    """tests/golden/config_proofs.json (whole proofs of the fib_endless run at 2^12 .. 2^20 cycles, written by make_config_proofs.py): the smallest entry is
    recomputed here by the oracle — length, SHA-256 and the spaced words — so the fixture the GPU test compares the 2^20-cycle proof with is known to be made
    the way its generator says; every entry names the same program, the shipped one."""
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_proofs.json")))
    blob = bytes.fromhex(gold["program_blob_hex"])
    assert blob == spec.fib_endless_program().to_bytes()
    assert {"12", "16", "18", "20"} <= set(gold["proofs"])
    g = gold["proofs"]["12"]
    res = oracle.run(blob, max_cycles=1 << 12, enable_execution_trace=True)
    pub = so.public_inputs(len(res.rows), blob, [], list(res.outputs), (res.halt_kind, res.halt_code))
    proof = np.ascontiguousarray(so.prove(res.rows, pub), dtype="<u4")
    assert len(proof) == g["words"] and hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
    pos = [int(i * (len(proof) - 1) // (len(g["samples"]) - 1)) for i in range(len(g["samples"]))]
    assert [int(proof[p]) for p in pos] == g["samples"]
    for k, e in list(gold["proofs"].items()) + list(gold["mode3_ring_proofs"].items()):
        assert e["rows"] == 1 << int(k) and len(e["samples"]) == 257 and len(e["sha256"]) == 64
    assert bytes.fromhex(gold["ring_program_blob_hex"]) == spec.memory_ring_program(10).to_bytes() and {"14", "16", "18"} <= set(gold["mode3_ring_proofs"])
