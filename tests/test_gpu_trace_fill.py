"""GPU parity: the HIP trace-fill path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from oracle import api as oracle
from zkir_amd import runtime as rt, spec

import helpers
import programs

pytestmark = pytest.mark.gpu


def _run_both(blob, inputs=(), **cfg):
    res = rt.VM(blob, inputs, rt.VMConfig(enable_execution_trace=True, **cfg)).run()
    want = oracle.run(blob, inputs, enable_execution_trace=True, **cfg)
    return res, want


def _check(res, want):
    assert res.cycles == want.cycles
    assert (res.halt_reason.kind, res.halt_reason.code) == (want.halt_kind, want.halt_code if want.halt_kind == 1 else 0)
    assert list(res.outputs) == list(want.outputs)
    helpers.assert_rows_equal(res.execution_trace.rows(), want.rows)


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 1023, 1024, 1025, 5000, 65536])
def test_fib_cycle_limit_sizes(n):
    """Ragged and exact tile multiples (default tile = 256 rows); halt = CycleLimit with exactly n rows (vm.rs:211-214)."""
    res, want = _run_both(spec.fib_endless_program().to_bytes(), max_cycles=n)
    assert res.cycles == n
    _check(res, want)
    res.close()


def test_fib_n30_exit():
    res, want = _run_both(spec.fib_program(30).to_bytes())
    assert res.outputs == [832040] and res.cycles == 154
    _check(res, want)
    res.close()


def test_empty_trace():
    res, want = _run_both(spec.fib_endless_program().to_bytes(), max_cycles=0)
    assert res.cycles == 0 and len(res.execution_trace) == 0
    _check(res, want)
    res.close()


@pytest.mark.parametrize("name", sorted(programs.ALL))
def test_program_suite(name):
    blob, inputs, cfg = programs.ALL[name]()
    res, want = _run_both(blob, inputs, **cfg)
    _check(res, want)
    res.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_programs(seed):
    blob, inputs = programs.random_program(seed, n_instr=400)
    for deferred in (False, True):
        try:
            want = oracle.run(blob, inputs, max_cycles=20000, enable_execution_trace=True, enable_deferred_model=deferred)
        except oracle.OracleError as e:
            with pytest.raises(rt.RuntimeError) as ei:
                rt.VM(blob, inputs, rt.VMConfig(max_cycles=20000, enable_execution_trace=True, enable_deferred_model=deferred)).run()
            assert ei.value.code == e.code
            continue
        res = rt.VM(blob, inputs, rt.VMConfig(max_cycles=20000, enable_execution_trace=True, enable_deferred_model=deferred)).run()
        _check(res, want)
        res.close()


def test_full_size_2p20_properties():
    """BASELINE configs[1] size (2^20 rows): bit-exact vs the oracle, plus size-independent properties."""
    n = 1 << 20
    blob = spec.fib_endless_program().to_bytes()
    res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()
    tr = res.execution_trace
    cyc = tr.column(rt.FIELD_CYCLE)
    assert np.array_equal(cyc, np.arange(n, dtype=np.uint64))
    # R0 is hard-wired zero with bound Constant(0) in every row
    assert not tr.column(rt.FIELD_REGISTERS, 0).any() and not tr.column(rt.FIELD_BOUND_BITS, 0).any()
    assert (tr.column(rt.FIELD_BOUND_TAG, 0) == 4).all()
    # fib recurrence mod 2^40 on the rows where `add r4, r1, r2` executes: next row's r4 = r1 + r2
    inst = tr.column(rt.FIELD_INSTRUCTION)
    r1, r2, r4 = (tr.column(rt.FIELD_REGISTERS, r) for r in (1, 2, 4))
    add_rows = np.nonzero(inst[:-1] == spec.add(4, 1, 2))[0]
    assert len(add_rows) > n // 6
    assert np.array_equal(r4[add_rows + 1], (r1[add_rows] + r2[add_rows]) & np.uint64((1 << 40) - 1))
    want = oracle.run(blob, max_cycles=n, enable_execution_trace=True)
    helpers.assert_rows_equal(tr.rows(), want.rows)
    res.close()


def test_sharded_fill_matches_unsharded():
    """Row sharding (multi-GPU path) on one device: two shard launches reproduce the unsharded rows."""
    import torch
    from zkir_amd import pipeline as pl
    n = 10 * 1024 + 300
    blob = spec.fib_endless_program().to_bytes()
    log = rt.interpret(blob, config=rt.VMConfig(max_cycles=n, enable_execution_trace=True))
    want = oracle.run(blob, max_cycles=n, enable_execution_trace=True).rows
    cut = 4 * 1024
    parts = []
    for lo, hi in ((0, cut), (cut, n)):
        sh = log.shard(lo, hi)
        ddl = pl.upload(sh)
        tr = pl.DeviceTrace(ddl)
        pl.trace_fill(pl.trace_fill_args(ddl, tr))
        torch.cuda.synchronize()
        parts.append(tr.rows())
        sh.close()
    helpers.assert_rows_equal(np.concatenate(parts), want)
    log.close()


# ---- zkir_exec's streaming path: runs with max_cycles >= 2^17 upload the finished part of the delta log and fill its tiles from a
# second host thread while the interpreter is still running --------------------------------------------------------------------
@pytest.mark.parametrize("n", [1 << 17, (1 << 17) + 5, 300_001])
def test_streaming_exec_cycle_limit(n):
    res, want = _run_both(spec.fib_endless_program().to_bytes(), max_cycles=n)
    assert res.cycles == n
    _check(res, want)
    res.close()


def test_streaming_exec_program_that_halts_early():
    """max_cycles = 10^6 (the default), the program exits after 327,684 rows: the device columns were sized for max_cycles."""
    blob = spec.fib_program(65536).to_bytes()
    res, want = _run_both(blob)
    assert res.cycles == 3 + 5 * 65535 + 6 and res.halt_reason == rt.HaltReason.Exit(0)
    _check(res, want)
    assert res.exec_stage_ms()["interpret"] > 0
    res.close()


def test_streaming_exec_abandoned_when_the_log_outgrows_its_reservation():
    """A loop whose XORs force a normalisation of both operands each time writes more than the 1.25 register events per row the
    streaming path reserves: the interpreter has to grow the event log, the uploader lets go of it and zkir_exec starts over on the
    plain path — same trace."""
    from zkir_amd.spec import Opcode as O, encode as E
    code = [spec.addi(1, 0, 5), spec.addi(2, 0, 9), spec.add(1, 1, 0), spec.add(2, 2, 0), E(O.XOR, 3, 1, 2),
            spec.add(1, 1, 0), spec.add(2, 2, 0), E(O.XOR, 3, 1, 2), spec.jal(0, -24)]
    res, want = _run_both(spec.Program.from_code(code).to_bytes(), max_cycles=1 << 18, enable_deferred_model=True)
    assert len(res.delta_log.reg_events) > 1.25 * (1 << 18)
    _check(res, want)
    res.close()


def test_streaming_exec_error_after_streaming_started():
    from zkir_amd.spec import Opcode as O, encode as E
    A = programs.A
    # 60000 iterations of a 4-instruction loop (240k rows), then a division by zero
    code = [A(3, 0, 30000), E(O.SLLI, 3, 3, imm=1), A(1, 1, 1), A(3, 3, -1), spec.bne(3, 0, -8), E(O.DIV, 2, 1, 0), programs.EB]
    blob = spec.Program.from_code(code).to_bytes()
    with pytest.raises(rt.RuntimeError) as e:
        rt.VM(blob, [], rt.VMConfig(enable_execution_trace=True)).run()
    assert e.value.code == rt.ERR_DIV_ZERO
    with pytest.raises(oracle.OracleError):
        oracle.run(blob, enable_execution_trace=True)
    res, want = _run_both(spec.fib_endless_program().to_bytes(), max_cycles=1 << 17)        # the library is fine afterwards
    _check(res, want)
    res.close()


@pytest.mark.parametrize("a,b", [(0, 700), (699, 1399), (1, 2), (255, 1025), (1398, 2100)])
def test_exec_shard_rows_and_witnesses_match_the_oracle_slice(a, b):
    """zkir_exec_shard: any row range of a finished interpretation as its own device trace — rows (absolute cycles), memory ops and
    SHA witnesses of the shard equal the oracle's for those rows."""
    blob = spec.sha256_chain_program().to_bytes()
    cfg = rt.VMConfig(max_cycles=2100, enable_execution_trace=True)
    log = rt.interpret(blob, [], cfg)
    want = oracle.run(blob, [], max_cycles=2100, enable_execution_trace=True)
    res = rt.exec_shard(log, a, b, blob, [], cfg)
    assert len(res.execution_trace) == b - a and res.delta_log.cycle_base == a
    helpers.assert_rows_equal(res.execution_trace.rows(), want.rows[a:b])
    ops, offs = res.row_memory_ops()
    lo, hi = int(want.row_memop_offsets[a]), int(want.row_memop_offsets[b])
    assert np.array_equal(ops, want.memops[lo:hi]) and np.array_equal(offs, want.row_memop_offsets[a:b + 1] - want.row_memop_offsets[a])
    res.close(); log.close()


def test_concurrent_zkir_exec_calls_from_several_threads():
    """The ABI is re-entrant: several host threads run zkir_exec at once (each streaming call has its own helper thread and a pooled
    upload stream; the log blocks come from a shared pool) — every result is the oracle's, bit for bit."""
    import threading
    jobs = [(spec.fib_endless_program().to_bytes(), dict(max_cycles=(1 << 17) + 11 * i)) for i in range(4)]
    jobs += [(spec.sha256_chain_program().to_bytes(), dict(max_cycles=150_000)), (spec.fib_program(2000).to_bytes(), {})]
    want = [oracle.run(b, [], enable_execution_trace=True, **kw) for b, kw in jobs]
    got, errs = [None] * len(jobs), []

    def work(i):
        try:
            b, kw = jobs[i]
            for _ in range(3):
                res = rt.VM(b, [], rt.VMConfig(enable_execution_trace=True, **kw)).run()
                rows = res.execution_trace.rows()
                outs, cyc = list(res.outputs), res.cycles
                res.close()
            got[i] = (rows, outs, cyc)
        except BaseException as e:                         # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for (rows, outs, cyc), w in zip(got, want):
        assert cyc == w.cycles and outs == list(w.outputs)
        helpers.assert_rows_equal(rows, w.rows)


# ---- zkir_exec_window: one GPU's share of a run executed on this rank (rows before it untraced, its own rows traced; streamed above 2^17 rows) ----
@pytest.mark.parametrize("a,b", [(0, 3000), (1000, 3000), (1, 2), (2999, 3000), (2500, 9000), (3000, 3001)])
def test_exec_window_rows_and_witnesses_match_the_oracle_slice(a, b):
    """The handle of a trace window — rows (absolute cycles), per-row memory ops and SHA witnesses — equals the oracle's rows [a, b) of
    the whole run; a window past the end of the run is cut at the halt."""
    blob = spec.sha256_chain_program().to_bytes()
    n = 3000
    want = oracle.run(blob, [], max_cycles=n, enable_execution_trace=True)
    res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run_window(a, b)
    lo, hi = min(a, n), min(b, n)
    assert len(res.execution_trace) == hi - lo and res.delta_log.cycle_base == lo and res.delta_log.window_open == (b < n)
    if hi > lo:
        helpers.assert_rows_equal(res.execution_trace.rows(), want.rows[lo:hi])
        ops, offs = res.row_memory_ops()
        o0, o1 = int(want.row_memop_offsets[lo]), int(want.row_memop_offsets[hi])
        assert np.array_equal(ops, want.memops[o0:o1]) and np.array_equal(offs, want.row_memop_offsets[lo:hi + 1] - want.row_memop_offsets[lo])
    res.close()


@pytest.mark.parametrize("a,b", [(1 << 17, 1 << 18), ((1 << 17) - 77, (1 << 18) + 1), (300_000, 300_000 + (1 << 17) + 3)])
def test_streaming_exec_window(a, b):
    """Windows of 2^17 rows and more take the streaming path (upload + K1 of the finished tiles under the running interpreter), with
    cycle_base = a: same rows as the oracle's slice of the whole run."""
    blob = spec.fib_endless_program().to_bytes()
    n = 300_000 + (1 << 17) + 3
    want = oracle.run(blob, [], max_cycles=n, enable_execution_trace=True, keep_rows=(a, b))
    res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run_window(a, b)
    assert len(res.execution_trace) == b - a and res.delta_log.cycle_base == a
    helpers.assert_rows_equal(res.execution_trace.rows(), want.rows)
    res.close()
