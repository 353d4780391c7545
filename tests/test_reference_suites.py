"""The reference's integration-test suites for the execution path (tests/golden/reference_suite_kats.json, transcribed by
tests/golden/make_suite_kats.py with file:line citations) against the oracle and the product's host interpreter (CPU), and — marked
gpu — through the C ABI's zkir_exec with every trace row compared with the oracle's.  An expectation is only ever something the
reference test itself asserts."""
import json
import os

import numpy as np
import pytest

from oracle import api as oracle
from zkir_amd import assembler, runtime as rt
from zkir_amd.spec import Program

import helpers

SUITE = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_suite_kats.json")))["programs"]
HALT = {"Ebreak": 0, "Exit": 1, "CycleLimit": 2}
IDS = [p["name"] for p in SUITE]


def _blob(p):
    prog = Program.from_code(p["code"])
    for k, v in p.get("program_config", {}).items():          # Program::with_config (zkir-spec/src/program.rs)
        setattr(prog, k, v)
    return prog.to_bytes()


def _run(p, impl):
    """-> dict(cycles, outputs, halt, n_rows, memops (row order), n_rc_w, rc_checks, norm) or the raised error code"""
    cfg = dict(p.get("config", {}))
    blob, inputs = _blob(p), p.get("inputs", [])
    if impl == "product_host":
        try:
            log = rt.interpret(blob, inputs, rt.VMConfig(**cfg))
        except rt.RuntimeError as e:
            return e.code, e.message
        return dict(cycles=log.cycles, outputs=list(log.outputs), halt=(log.halt_reason.kind, log.halt_reason.code), n_rows=log.n_rows,
                    memops=helpers.memops_from_log(log), n_rc_w=len(log.rc_offsets) - 1, rc_checks=helpers.rc_from_log(log), norm=helpers.norm_from_log(log),
                    rows=helpers.expand_delta_log(log) if log.n_rows else None)
    try:
        r = oracle.run(blob, inputs, **cfg)
    except oracle.OracleError as e:
        return e.code, e.msg
    return dict(cycles=r.cycles, outputs=list(r.outputs), halt=(r.halt_kind, r.halt_code if r.halt_kind == 1 else 0), n_rows=len(r.rows), memops=r.memops,
                n_rc_w=len(r.rc_offsets) - 1, rc_checks=r.rc_checks, norm=r.norm_events, rows=r.rows)


def _witness_verifies(e):
    """NormalizationWitness::verify (zkir-runtime/src/normalization_witness.rs:82-103)"""
    nb = int(e["normalized_bits"]); mask = (1 << nb) - 1
    a0, a1 = int(e["accumulated"][0]), int(e["accumulated"][1])
    if int(e["carries"][0]) != a0 >> nb or int(e["normalized"][0]) != a0 & mask:
        return False
    t = a1 + int(e["carries"][0])
    return int(e["carries"][1]) == t >> nb and int(e["normalized"][1]) == t & mask


def _check(p, r):
    if "error" in p:
        assert isinstance(r, tuple) and r[0] == p["error"], r
        if "error_message" in p:                              # the reference's Display text (zkir-runtime/src/error.rs)
            assert r[1] == p["error_message"], r
        return
    assert isinstance(r, dict), f"run failed with error {r}"
    if "outputs" in p:
        assert r["outputs"] == p["outputs"]
    if "cycles" in p:
        assert r["cycles"] == p["cycles"]
    if "cycles_min" in p:
        assert r["cycles"] >= p["cycles_min"]
    if p.get("ok"):
        assert r["cycles"] > 0
    if "halt" in p:
        assert r["halt"] == (HALT[p["halt"][0]], p["halt"][1] if len(p["halt"]) > 1 else 0)
    if "n_rows" in p:
        assert r["n_rows"] == p["n_rows"]
    if "n_rows_min" in p:
        assert r["n_rows"] >= p["n_rows_min"]
    if "final_bound" in p:                                    # pre-state of the last row = after everything but the halting instruction
        assert int(r["rows"][-1]["bound_bits"][p["final_bound"]["reg"]]) == p["final_bound"]["bits"]
    for reg, bits in p.get("final_bounds", {}).items():
        assert int(r["rows"][-1]["bound_bits"][int(reg)]) == bits, (reg, int(r["rows"][-1]["bound_bits"][int(reg)]), bits)
    if "n_memops" in p:
        assert len(r["memops"]) == p["n_memops"]
    for got, want in zip(r["memops"], p.get("memops", [])):
        for k, v in want.items():
            assert int(got[k]) == v, (k, int(got[k]), v)
    if "rc_witnesses" in p:
        assert r["n_rc_w"] == p["rc_witnesses"]
    if "rc_witnesses_min" in p:
        assert r["n_rc_w"] >= p["rc_witnesses_min"]
    if "rc_checks_total" in p:
        assert len(r["rc_checks"]) == p["rc_checks_total"]
    if "rc_check_pcs" in p:
        assert [int(x) for x in r["rc_checks"]["pc"]] == p["rc_check_pcs"]
    for c in r["rc_checks"]:                                  # range_checking.rs:232-259: chunks below 2^10 that recompose the 40-bit value
        ch = [int(x) for x in c["chunks"]]
        assert all(x < 1024 for x in ch)
        assert (ch[0] | ch[1] << 10) | (ch[2] | ch[3] << 10) << 20 == int(c["value"]) & ((1 << 40) - 1)
    norm = r["norm"]
    if "norm_events" in p:
        assert len(norm) == p["norm_events"]
    if "norm_events_min" in p:
        assert len(norm) >= p["norm_events_min"]
    if p.get("norm_all_observation"):
        assert (norm["cause"] == 0).all()                      # NormalizationCause::ObservationPoint, with a triggering opcode
    if p.get("norm_all_verify"):
        assert all(_witness_verifies(e) for e in norm)
    if p.get("norm_any_carry"):
        assert any(int(e["carries"][0]) or int(e["carries"][1]) for e in norm)
    if "norm_min_cycle" in p:
        assert (norm["cycle"] >= p["norm_min_cycle"]).all() and (norm["pc"] >= p["norm_min_pc"]).all()
    if "norm_event_for" in p:
        w = p["norm_event_for"]
        hits = [e for e in norm if int(e["reg"]) == w["register"]]
        assert hits and [int(x) for x in hits[-1]["normalized"]] == w["normalized"] and [int(x) for x in hits[-1]["carries"]] == w["carries"]


@pytest.mark.parametrize("p", SUITE, ids=IDS)
@pytest.mark.parametrize("impl", ["oracle", "product_host"])
def test_reference_suite(p, impl):
    _check(p, _run(p, impl))


@pytest.mark.parametrize("p", [p for p in SUITE if "source" in p], ids=[p["name"] for p in SUITE if "source" in p])
def test_reference_sources_assemble_to_the_transcribed_words(p):
    """N2: the product's assembler on the reference's own test sources gives the words the fixture generator encoded by hand."""
    prog = assembler.assemble(p["source"])
    assert list(prog.code) == p["code"]


def test_host_and_oracle_agree_row_for_row_on_every_suite_program():
    """Beyond what the reference asserts: the product's host log expands to exactly the oracle's rows, for every suite program."""
    for p in SUITE:
        if "error" in p:
            continue
        cfg = dict(p.get("config", {})); cfg["enable_execution_trace"] = True
        blob, inputs = _blob(p), p.get("inputs", [])
        want = oracle.run(blob, inputs, **cfg)
        log = rt.interpret(blob, inputs, rt.VMConfig(**cfg))
        helpers.assert_rows_equal(helpers.expand_delta_log(log), want.rows)
        assert list(log.outputs) == list(want.outputs) and log.cycles == want.cycles


@pytest.mark.gpu
def test_reference_suite_through_zkir_exec_on_the_gpu():
    """Every suite program through the drop-in call on the device: expectations of the reference + every row bit-exact vs the oracle."""
    for p in SUITE:
        cfg = dict(p.get("config", {})); cfg["enable_execution_trace"] = True
        blob, inputs = _blob(p), p.get("inputs", [])
        if "error" in p:
            with pytest.raises(rt.RuntimeError) as e:
                rt.VM(blob, inputs, rt.VMConfig(**cfg)).run()
            assert e.value.code == p["error"], p["name"]
            continue
        res = rt.VM(blob, inputs, rt.VMConfig(**cfg)).run()
        want = oracle.run(blob, inputs, **cfg)
        assert res.cycles == want.cycles and list(res.outputs) == list(want.outputs), p["name"]
        if "outputs" in p:
            assert list(res.outputs) == p["outputs"], p["name"]
        if "halt" in p:
            assert (res.halt_reason.kind, res.halt_reason.code) == (HALT[p["halt"][0]], p["halt"][1] if len(p["halt"]) > 1 else 0), p["name"]
        helpers.assert_rows_equal(res.execution_trace.rows(), want.rows)
        # the rest of ExecutionResult, expanded on the device behind the same handle (vm.rs:54-103)
        ops, offs = res.row_memory_ops()
        assert np.array_equal(ops, want.memops) and np.array_equal(offs, want.row_memop_offsets), p["name"]
        assert np.array_equal(res.get_memory_trace(), want.sorted_memops), p["name"]
        flat = [c for grp in res.range_check_witnesses for c in grp]
        assert flat == [(int(e["value"]), [int(x) for x in e["chunks"]], int(e["pc"])) for e in want.rc_checks], p["name"]
        assert np.array_equal(res.normalization_witnesses, want.norm_events), p["name"]
        res.close()
