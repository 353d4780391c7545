"""GPU parity for the witness-expansion kernels (witness.hip) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from oracle import api as oracle
from zkir_amd import runtime as rt, spec

import programs

pytestmark = pytest.mark.gpu


def _both(blob, inputs, cfg):
    cfg = dict(cfg, enable_execution_trace=True)
    log = rt.interpret(blob, inputs, rt.VMConfig(**cfg))
    want = oracle.run(blob, inputs, **cfg)
    return log, want


MEM_PROGRAMS = ["mem_sw_lw", "timestamps", "loads_stores", "q9_access_at_own_pc", "hashes_all", "blake3_multi_chunk", "sha_chain_small",
                "sha256_hello", "deferred_negative_and_overflow", "fib30"]


@pytest.mark.parametrize("separate", [False, True], ids=["one_pass_csr", "separate_passes"])
@pytest.mark.parametrize("name", MEM_PROGRAMS)
def test_memory_ops_row_order_offsets_and_sorted(name, separate):
    """Both ways to the same columns: the single-pass expansion that also emits the CSR offsets and the shape flags (what the result
    handle uses), and the stand-alone entry points (binary-search CSR, expansion, sort with its own check pass)."""
    from zkir_amd import pipeline as pl
    blob, inputs, cfg = programs.ALL[name]()
    log, want = _both(blob, inputs, cfg)
    rows, offsets, srt = pl.memory_ops(log, separate_passes=separate)
    assert np.array_equal(rows.to_numpy(), want.memops)
    assert np.array_equal(offsets.cpu().numpy().view(np.uint64), want.row_memop_offsets)
    assert np.array_equal(srt.to_numpy(), want.sorted_memops)            # ExecutionResult::get_memory_trace, vm.rs:85-94


def test_memory_ops_csr_with_long_gaps_between_ops():
    """A program that touches memory rarely: thousands of rows without an op between two ops, ops in the first and the last row, a
    tail of rows after the last op — the single-pass CSR fills every gap (short ones by the lane that sees the boundary, long ones
    by its workgroup)."""
    from zkir_amd import pipeline as pl
    from zkir_amd.spec import Opcode as O, encode as E
    A = programs.A
    loop = lambda cnt: [A(3, 0, cnt), A(1, 1, 1), A(3, 3, -1), spec.bne(3, 0, -8)]          # noqa: E731 - 1 + 3 cnt rows without memory ops
    code = ([E(O.SW, rs1=0, rs2=0, imm=0x4000)] + loop(5) + [A(5, 0, 0x2000), E(O.SW, rs1=5, rs2=1, imm=0)] + loop(3000) + [E(O.LW, 2, 5, imm=0)] + loop(20)
            + [E(O.SW, rs1=5, rs2=1, imm=4), E(O.LW, 2, 5, imm=4)] + loop(700) + [E(O.LW, 2, 5, imm=0)])
    for tail in ([programs.EB], loop(9000) + [programs.EB]):
        blob = spec.Program.from_code(code + tail).to_bytes()
        log, want = _both(blob, [], {})
        assert len(want.memops) == 6 and want.memops["timestamp"][0] == 0
        for separate in (False, True):
            rows, offsets, srt = pl.memory_ops(log, separate_passes=separate)
            assert np.array_equal(rows.to_numpy(), want.memops)
            assert np.array_equal(offsets.cpu().numpy().view(np.uint64), want.row_memop_offsets)
            assert np.array_equal(srt.to_numpy(), want.sorted_memops)
    # no memory ops at all: every offset is zero
    log, want = _both(spec.fib_endless_program().to_bytes(), [], dict(max_cycles=5000))
    rows, offsets, srt = pl.memory_ops(log)
    assert len(want.memops) == 0 and np.array_equal(offsets.cpu().numpy().view(np.uint64), want.row_memop_offsets)


def test_memory_sort_fallback_on_wrapping_addresses():
    """A hash input that wraps around 2^64 breaks the ascending-run shape: the counting-rank path must still match."""
    from zkir_amd import pipeline as pl
    from zkir_amd.spec import Opcode as O, encode as E
    A = programs.A
    # LB keeps the 64-bit sign extension (Q1), which is the only way to get a raw register value near 2^64
    code = [A(5, 0, 0x2000), A(9, 0, 0xF8), E(O.SB, rs1=5, rs2=9, imm=0), A(9, 0, 0xF0), E(O.SB, rs1=5, rs2=9, imm=1), A(9, 0, 0xFD),
            E(O.SB, rs1=5, rs2=9, imm=2),
            E(O.LB, 11, 5, imm=0), A(12, 0, 20), A(13, 0, 0x3000), A(10, 0, 5), programs.EC,            # keccak over [2^64-8, 2^64+12)
            E(O.LB, 11, 5, imm=2), A(12, 0, 9), E(O.LB, 13, 5, imm=1), A(10, 0, 6), programs.EC,          # blake3, output wraps too
            programs.EB]
    blob = spec.Program.from_code(code).to_bytes()
    log, want = _both(blob, [], {})
    rows, offsets, srt = pl.memory_ops(log)
    assert len(want.memops) == 6 + 20 + 32 + 9 + 32
    assert want.memops["address"].max() > 2**63 and want.memops["address"].min() == 0     # the syscall rows really wrap
    assert np.array_equal(rows.to_numpy(), want.memops)
    assert np.array_equal(srt.to_numpy(), want.sorted_memops)


@pytest.mark.parametrize("seed", range(6))
def test_random_program_witnesses(seed):
    from zkir_amd import pipeline as pl
    blob, inputs = programs.random_program(100 + seed, n_instr=500)
    cfg = dict(max_cycles=20000, enable_range_checking=True, enable_deferred_model=bool(seed % 2))
    try:
        log, want = _both(blob, inputs, cfg)
    except (oracle.OracleError, rt.RuntimeError):
        pytest.skip("program errors out")
    rows, offsets, srt = pl.memory_ops(log)
    assert np.array_equal(rows.to_numpy(), want.memops) and np.array_equal(srt.to_numpy(), want.sorted_memops)
    value, pc, chunks, mult = pl.range_checks(log)
    assert np.array_equal(value.cpu().numpy().view(np.uint64), want.rc_checks["value"])
    assert np.array_equal(pc.cpu().numpy().view(np.uint64), want.rc_checks["pc"])
    assert np.array_equal(chunks.cpu().numpy().view(np.uint16).T, want.rc_checks["chunks"])
    assert np.array_equal(pl.normalization_events(log), want.norm_events)


@pytest.mark.parametrize("name", ["rc_doubling", "rc_many_pending", "rc_config_30bit", "fib_rc"])
def test_range_check_chunks_and_multiplicities(name):
    from zkir_amd import pipeline as pl
    blob, inputs, cfg = programs.ALL[name]()
    log, want = _both(blob, inputs, cfg)
    assert len(want.rc_checks) > 0
    value, pc, chunks, mult = pl.range_checks(log)
    assert np.array_equal(value.cpu().numpy().view(np.uint64), want.rc_checks["value"])
    assert np.array_equal(pc.cpu().numpy().view(np.uint64), want.rc_checks["pc"])
    assert np.array_equal(chunks.cpu().numpy().view(np.uint16).T, want.rc_checks["chunks"])
    assert np.array_equal(log.rc_offsets, want.rc_offsets)
    table = 1 << log.rc_chunk_bits
    assert np.array_equal(mult.cpu().numpy().view(np.uint32), np.bincount(want.rc_checks["chunks"].reshape(-1), minlength=table).astype(np.uint32))


@pytest.mark.parametrize("name", ["deferred_add_branch", "deferred_chain_store", "deferred_add_sub_mix", "deferred_negative_and_overflow", "deferred_fib"])
def test_normalization_events(name):
    from zkir_amd import pipeline as pl
    blob, inputs, cfg = programs.ALL[name]()
    log, want = _both(blob, inputs, cfg)
    # deferred_chain_store stores through rs1 = R0: norm_two! emits nothing for R0 and normalises rs2 silently (execute.rs:903-916)
    assert len(want.norm_events) > 0 or name == "deferred_chain_store"
    assert np.array_equal(pl.normalization_events(log), want.norm_events)


@pytest.mark.parametrize("name", MEM_PROGRAMS + ["rc_doubling", "rc_many_pending", "rc_config_30bit", "fib_rc", "deferred_fib", "deferred_add_sub_mix"])
def test_result_handle_witness_streams(name):
    """The drop-in handle itself (zkir_exec -> zkir_result_memory_trace / _range_check_witnesses / _normalization_witnesses /
    _sha256_witnesses): device columns behind the C ABI, copied back and compared with the oracle's ExecutionResult."""
    blob, inputs, cfg = programs.ALL[name]()
    cfg = dict(cfg, enable_execution_trace=True)
    res = rt.VM(blob, inputs, rt.VMConfig(**cfg)).run()
    want = oracle.run(blob, inputs, **cfg)
    ops, offs = res.row_memory_ops()
    assert np.array_equal(ops, want.memops) and np.array_equal(offs, want.row_memop_offsets)
    assert np.array_equal(res.get_memory_trace(), want.sorted_memops) and res.memory_op_count() == len(want.memops)
    assert np.array_equal(res.get_memory_trace(), want.sorted_memops)                # cached columns: same answer the second time
    w = res.range_check_witnesses
    assert [len(x) for x in w] == list(np.diff(want.rc_offsets.astype(np.int64)))
    flat = [c for grp in w for c in grp]
    assert flat == [(int(e["value"]), [int(x) for x in e["chunks"]], int(e["pc"])) for e in want.rc_checks]
    if len(want.rc_checks):
        rcw = res.range_check_witness()
        mult = res._d2h(rcw.multiplicity, 1 << rcw.chunk_bits, "<u4")
        assert np.array_equal(mult, np.bincount(want.rc_checks["chunks"].reshape(-1), minlength=1 << rcw.chunk_bits).astype(np.uint32))
    assert np.array_equal(res.normalization_witnesses, want.norm_events)
    cols, stamps = res.sha256_witnesses()
    blocks = res.delta_log.sha_blocks
    assert cols.shape[1] == len(blocks) == len(stamps)
    for k in range(len(blocks)):
        n_bytes = int(blocks[k]["message_block"][15]) // 8
        msg = blocks[k]["message_block"].astype(">u4").tobytes()[:n_bytes]
        assert np.array_equal(cols[:, k], oracle.sha256_witness(msg, int(stamps[k]))["flat"])
    res.close()


def test_execution_result_members_mirror_the_reference():
    """ExecutionResult.{get_memory_trace, memory_op_count, range_check_witnesses, normalization_witnesses} (vm.rs:54-103)."""
    blob, inputs, cfg = programs.ALL["deferred_negative_and_overflow"]()
    cfg = dict(cfg, enable_range_checking=True, enable_execution_trace=True)
    res = rt.VM(blob, inputs, rt.VMConfig(**cfg)).run()
    want = oracle.run(blob, inputs, **cfg)
    assert np.array_equal(res.get_memory_trace(), want.sorted_memops) and res.memory_op_count() == len(want.memops)
    assert np.array_equal(res.normalization_witnesses, want.norm_events)
    res.close()
    blob, inputs, cfg = programs.ALL["rc_many_pending"]()
    res = rt.VM(blob, inputs, rt.VMConfig(enable_execution_trace=True, **cfg)).run()
    want = oracle.run(blob, inputs, enable_execution_trace=True, **cfg)
    w = res.range_check_witnesses
    assert len(w) == len(want.rc_offsets) - 1 and sum(len(x) for x in w) == len(want.rc_checks)
    flat = [c for grp in w for c in grp]
    for got, exp in zip(flat, want.rc_checks):
        assert got == (int(exp["value"]), [int(x) for x in exp["chunks"]], int(exp["pc"]))
    res.close()


def test_sha256_chip_matches_witness():
    """K3 vs sha256_hash_with_witness (crypto.rs:223-297): all 608 words per block, every single-block length 0..55."""
    from zkir_amd import pipeline as pl
    msgs = [bytes((11 * i + n) & 0xFF for i in range(n)) for n in range(56)] + [b"hello", b"", b"abc"]
    blocks = np.zeros(len(msgs), dtype=rt.SHA_BLOCK_DTYPE)
    want = np.zeros((len(msgs), 608), dtype=np.uint32)
    for k, m in enumerate(msgs):
        w = oracle.sha256_witness(m, 1000 + k)
        blocks[k]["message_block"] = w["message_block"]
        blocks[k]["timestamp"] = 1000 + k
        want[k] = w["flat"]
    cols, ts = pl.sha256_chip(blocks)
    assert np.array_equal(cols.cpu().numpy().view(np.uint32).T, want)
    assert np.array_equal(ts.cpu().numpy(), 1000 + np.arange(len(msgs)))


def test_sha256_chip_from_hash_chain_run():
    """Config 5 shape: the blocks recorded by the host for a SHA-256 hash chain, expanded on the device."""
    from zkir_amd import pipeline as pl
    import hashlib
    blob = spec.sha256_chain_program().to_bytes()
    log = rt.interpret(blob, config=rt.VMConfig(max_cycles=1 << 14, enable_execution_trace=True))
    assert len(log.sha_blocks) > 2000
    cols, ts = pl.sha256_chip(log.sha_blocks)
    got = cols.cpu().numpy().view(np.uint32)
    for k in (0, 1, 777, len(log.sha_blocks) - 1):
        blk = log.sha_blocks[k]["message_block"].astype(">u4").tobytes()[:32]      # 32-byte message
        w = oracle.sha256_witness(blk, int(log.sha_blocks[k]["timestamp"]))
        assert np.array_equal(got[:, k], w["flat"])
        assert got[600:608, k].astype(">u4").tobytes() == hashlib.sha256(blk).digest()
