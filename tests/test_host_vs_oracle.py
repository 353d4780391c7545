"""The product's host interpreter (delta log, expanded in numpy by tests/helpers.py) against the CPU oracle.
Covers everything the reference's VM::run produces: rows, memory ops (row order and get_memory_trace order),
outputs, halt reason, cycle count, range-check witnesses, normalization events, SHA blocks, error kinds.
No GPU needed: this validates the host half of the product; the HIP half is checked in test_gpu_*.py."""
import numpy as np
import pytest

from oracle import api as oracle
from zkir_amd import runtime as rt, spec

import helpers
import programs


def compare(blob, inputs, cfg, tile_rows=0):
    cfg = dict(cfg)
    cfg.setdefault("enable_execution_trace", True)
    want = oracle.run(blob, inputs, **cfg)
    log = rt.interpret(blob, inputs, rt.VMConfig(**cfg), tile_rows=tile_rows)
    assert log.cycles == want.cycles
    assert (log.halt_reason.kind, log.halt_reason.code) == (want.halt_kind, want.halt_code if want.halt_kind == 1 else 0)
    assert list(log.outputs) == list(want.outputs)
    assert log.n_rows == len(want.rows)
    helpers.check_tile_index(log)
    helpers.assert_rows_equal(helpers.expand_delta_log(log), want.rows)
    got_mem = helpers.memops_from_log(log)
    assert np.array_equal(got_mem, want.memops), "memory ops (row order) differ"
    if log.n_rows:
        offs = np.searchsorted(log.mem_events["row"], np.arange(log.n_rows + 1))
        assert np.array_equal(offs.astype(np.uint64), want.row_memop_offsets)
    assert np.array_equal(helpers.rc_from_log(log), want.rc_checks)
    assert np.array_equal(log.rc_offsets, want.rc_offsets)
    assert np.array_equal(helpers.norm_from_log(log), want.norm_events)
    return log, want


@pytest.mark.parametrize("name", sorted(programs.ALL))
def test_program_suite(name):
    blob, inputs, cfg = programs.ALL[name]()
    compare(blob, inputs, cfg)


@pytest.mark.parametrize("name", sorted(programs.ALL))
def test_program_suite_all_modes(name):
    """Every program again with range checking + deferred model forced on, and with the faithful O(N^2) oracle."""
    blob, inputs, cfg = programs.ALL[name]()
    cfg = dict(cfg, enable_range_checking=True, enable_deferred_model=True)
    try:
        compare(blob, inputs, cfg, tile_rows=256)
    except oracle.OracleError as e:
        with pytest.raises(rt.RuntimeError) as ei:
            rt.interpret(blob, inputs, rt.VMConfig(enable_execution_trace=True, **cfg))
        assert ei.value.code == e.code
    cfg2 = dict(programs.ALL[name]()[2], enable_execution_trace=True)
    a = oracle.run(blob, inputs, faithful=True, **cfg2)
    b = oracle.run(blob, inputs, faithful=False, **cfg2)
    assert np.array_equal(a.rows, b.rows) and np.array_equal(a.memops, b.memops) and np.array_equal(a.row_memop_offsets, b.row_memop_offsets)


@pytest.mark.parametrize("name", sorted(programs.ERRORS))
def test_error_kinds(name):
    blob, inputs, cfg, code = programs.ERRORS[name]()
    with pytest.raises(oracle.OracleError) as eo:
        oracle.run(blob, inputs, enable_execution_trace=True, **cfg)
    with pytest.raises(rt.RuntimeError) as ep:
        rt.interpret(blob, inputs, rt.VMConfig(enable_execution_trace=True, **cfg))
    assert eo.value.code == code and ep.value.code == code
    assert ep.value.message == eo.value.msg, (ep.value.message, eo.value.msg)


@pytest.mark.parametrize("seed", range(40))
def test_random_programs(seed):
    blob, inputs = programs.random_program(seed, n_instr=300 + 20 * seed)
    for deferred, rc in ((False, False), (True, True), (False, True)):
        cfg = dict(max_cycles=5000, enable_deferred_model=deferred, enable_range_checking=rc)
        try:
            compare(blob, inputs, cfg, tile_rows=256 if seed % 2 else 0)
        except oracle.OracleError as e:
            with pytest.raises(rt.RuntimeError) as ei:
                rt.interpret(blob, inputs, rt.VMConfig(enable_execution_trace=True, **cfg))
            assert ei.value.code == e.code and ei.value.message == e.msg


def test_trace_disabled_has_no_rows_or_memops():
    blob, inputs, cfg = programs.ALL["hashes_all"]()
    log = rt.interpret(blob, inputs, rt.VMConfig(**cfg))
    want = oracle.run(blob, inputs, **cfg)
    assert log.n_rows == 0 and len(log.mem_events) == 0 and len(log.reg_events) == 0 and log.cycles == want.cycles
    assert list(log.outputs) == list(want.outputs)


def test_sha_blocks_match_witness_inputs():
    """The SHA-chip input blocks the host records are exactly parse_message_block(pad_message(input)) (crypto.rs:108-139)."""
    blob = spec.sha256_chain_program().to_bytes()
    log = rt.interpret(blob, config=rt.VMConfig(max_cycles=400, enable_execution_trace=True))
    want = oracle.run(blob, max_cycles=400, enable_execution_trace=True)
    assert len(log.sha_blocks) == (400 - 13 + 4) // 6 or len(log.sha_blocks) > 50
    seed = bytes(range(32))
    msg = seed
    import hashlib
    for blk in log.sha_blocks[:20]:
        w = oracle.sha256_witness(msg, int(blk["timestamp"]))
        assert np.array_equal(blk["message_block"], w["message_block"])
        # the ECALL row's memory ops hold the digest written by that syscall
        lo, hi = int(want.row_memop_offsets[blk["timestamp"]]), int(want.row_memop_offsets[blk["timestamp"] + 1])
        ops = want.memops[lo:hi]
        assert len(ops) == 40 and ops["is_write"][:32].sum() == 0 and ops["is_write"][32:].all()
        digest = hashlib.sha256(msg).digest()
        assert ops["value"][32:].astype(">u4").tobytes() == digest
        msg = ops["value"][32:].astype("<u4").tobytes()     # BE-parsed words stored with LE write_u32 (crypto.rs:252-255): memory holds byte-swapped words


def test_sorted_memory_trace_is_stable_by_ts_addr_rw():
    """ExecutionResult::get_memory_trace (vm.rs:85-94): stable sort by (timestamp, address, Read<Write)."""
    blob, inputs, cfg = programs.ALL["hashes_all"]()
    want = oracle.run(blob, inputs, enable_execution_trace=True, **cfg)
    ops = want.memops
    order = np.lexsort((np.arange(len(ops)), ops["is_write"], ops["address"], ops["timestamp"]))
    assert np.array_equal(ops[order], want.sorted_memops)


def test_shard_equals_slice():
    n = 7 * 256 + 77
    blob = spec.sha256_chain_program().to_bytes()
    cfg = rt.VMConfig(max_cycles=n, enable_execution_trace=True)
    log = rt.interpret(blob, config=cfg, tile_rows=256)
    full = helpers.expand_delta_log(log)
    for lo, hi in ((0, 512), (512, 1024), (1024, n), (256, n), (0, n), (768, 768)):
        sh = log.shard(lo, hi)
        helpers.check_tile_index(sh)
        rows = helpers.expand_delta_log(sh)
        rows["cycle"] += np.uint64(sh.cycle_base)
        helpers.assert_rows_equal(rows, full[lo:hi])
        me = log.mem_events
        sel = me[(me["row"] >= lo) & (me["row"] < hi)].copy()
        sel["row"] -= lo
        assert np.array_equal(sh.mem_events, sel)
        sh.close()
    sh = log.shard(100, 200)         # not tile-aligned: the shard gets its own tiling (tests/test_shard_logs.py)
    rows = helpers.expand_delta_log(sh)
    rows["cycle"] += np.uint64(100)
    helpers.assert_rows_equal(rows, full[100:200])
    sh.close()
    with pytest.raises(rt.RuntimeError):
        log.shard(200, 100)          # empty-negative range


def test_fib_2p16_faithful_equals_linear():
    blob = spec.fib_endless_program().to_bytes()
    a = oracle.run(blob, max_cycles=1 << 13, enable_execution_trace=True, faithful=True)
    log, b = compare(blob, [], dict(max_cycles=1 << 13))
    assert np.array_equal(a.rows, b.rows)
    # bounds never tighten in this loop: max_bits of r1/r2/r4 only grow
    assert (np.diff(b.rows["bound_bits"][8:, 4].astype(np.int64)) >= 0).all() and b.rows["bound_bits"][-1, 4] > 40


def test_program_blob_validation_matches_the_oracle_on_mutated_headers():
    """ProgramHeader::from_bytes / validate and Program::from_bytes (zkir-spec/src/program.rs:147-214, :318-346): every header field
    pushed off its valid range, every kind of truncation, and 300 random single-byte corruptions of a valid blob — the product's
    loader gives the oracle's verdict: the same code and the same message, or the same run."""
    good = bytearray(spec.fib_program(7).to_bytes())

    def same(blob):
        blob = bytes(blob)
        try:
            want = oracle.run(blob, [], max_cycles=2000, enable_execution_trace=True)
        except oracle.OracleError as e:
            with pytest.raises(rt.RuntimeError) as ei:
                rt.interpret(blob, [], rt.VMConfig(max_cycles=2000, enable_execution_trace=True))
            assert (ei.value.code, ei.value.message) == (e.code, e.msg), (blob[:32].hex(), ei.value.message, e.msg)
            return e.code
        log = rt.interpret(blob, [], rt.VMConfig(max_cycles=2000, enable_execution_trace=True))
        assert log.cycles == want.cycles and list(log.outputs) == list(want.outputs)
        return 0

    assert same(good) == 0
    def with_(off, val, width=1):
        b = bytearray(good); b[off:off + width] = int(val).to_bytes(width, "little"); return b
    rejected = 0
    for b in [with_(0, 0x5A4B4953, 4), with_(4, 0x00030003, 4), with_(4, 0x00040000, 4)]:        # magic, version (program.rs:37,40,147-160)
        rejected += same(b) == 7
    for limb_bits in (0, 15, 17, 21, 31, 32, 255):                                                # config.rs:154-174
        rejected += same(with_(8, limb_bits)) == 7
    for data_limbs in (0, 5, 255):
        rejected += same(with_(9, data_limbs)) == 7
    for addr_limbs in (0, 3, 255):
        rejected += same(with_(10, addr_limbs)) != 0
    assert rejected >= 15
    for limb_bits, data_limbs in ((16, 1), (18, 2), (30, 3), (24, 4)):                             # valid non-default configurations run the same on both sides
        b = with_(8, limb_bits); b[9] = data_limbs
        same(b)
    for cut in (0, 1, 4, 31, 32, 33, len(good) - 1, len(good) - 4):                                # truncations
        same(good[:cut])
    same(good + b"\\x00" * 7)                                                                       # trailing bytes
    for off, width in ((12, 4), (16, 4), (20, 4), (24, 4), (28, 4)):                               # entry point and the four section sizes: small and huge values
        for val in (0, 1, 4, 0x1000, 0x1004, len(good), 0x7FFFFFFF, 0xFFFFFFFF):
            same(with_(off, val, width))
    rng = np.random.default_rng(99)
    for _ in range(300):
        b = bytearray(good)
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        same(b)
