"""Row sharding of the delta log (zkir_delta_log_shard, multi-GPU path): the side logs of the shards — memory ops, range-check
witnesses, normalization events, SHA blocks — concatenated in shard order equal the unsharded ones, and every shard's register
events expand to exactly its slice of the rows.  Host only (no GPU)."""
import numpy as np
import pytest

from zkir_amd import runtime as rt, spec

import helpers
import programs


def _cases():
    yield "fib_rc_long", spec.fib_program(400).to_bytes(), [], dict(enable_range_checking=True)
    yield "sha_chain", spec.sha256_chain_program().to_bytes(), [], dict(max_cycles=3000)
    yield "deferred_fib_rc", spec.fib_program(300).to_bytes(), [], dict(enable_deferred_model=True, enable_range_checking=True)
    for seed in (3, 4, 5):
        blob, inputs = programs.random_program(200 + seed, n_instr=400, range_checking=True)
        yield f"random{seed}", blob, inputs, dict(max_cycles=6000, enable_range_checking=True, enable_deferred_model=bool(seed & 1))


_CASES = {c[0]: c[1:] for c in _cases()}


@pytest.mark.parametrize("name", sorted(_CASES))
def test_shards_concatenate_to_the_whole(name):
    blob, inputs, cfg = _CASES[name]
    try:
        log = rt.interpret(blob, inputs, rt.VMConfig(enable_execution_trace=True, **cfg), tile_rows=256)
    except rt.RuntimeError:
        pytest.skip("program errors out")
    n, T = log.n_rows, 256
    assert n > T
    cuts = [0] + [c for c in (T, 3 * T) if c < n] + [n]
    whole_rows = helpers.expand_delta_log(log)
    mem, rc_ev, rc_sizes, rc_cyc, norm, sha = [], [], [], [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        sh = log.shard(a, b)
        assert sh.n_rows == b - a and sh.cycle_base == a
        rows = helpers.expand_delta_log(sh)
        rows["cycle"] += np.uint64(a)
        helpers.assert_rows_equal(rows, whole_rows[a:b])
        m = sh.mem_events.copy(); m["row"] += np.uint32(a); mem.append(m)
        rc_ev.append(sh.rc_events.copy()); rc_sizes += list(np.diff(sh.rc_offsets.astype(np.int64))); rc_cyc += list(sh.rc_cycles)
        assert all(a <= c < b for c in sh.rc_cycles)
        norm.append(sh.norm_events.copy()); sha.append(sh.sha_blocks.copy())
        sh.close()
    assert np.array_equal(np.concatenate(mem), log.mem_events)
    assert np.array_equal(np.concatenate(rc_ev), log.rc_events)
    assert rc_sizes == list(np.diff(log.rc_offsets.astype(np.int64))) and rc_cyc == list(log.rc_cycles)
    assert np.array_equal(np.concatenate(norm), log.norm_events)
    assert np.array_equal(np.concatenate(sha), log.sha_blocks)
    if name == "fib_rc_long":
        assert len(log.rc_events) > 0 and len(log.rc_cycles) == len(log.rc_offsets) - 1
    log.close()


@pytest.mark.parametrize("name", sorted(_CASES))
def test_unaligned_and_overlapping_shards_expand_to_their_rows(name):
    """Cuts that are not on tile boundaries (segment proofs overlap by one row): the shard's own tiling — snapshot at its first row,
    tile offsets and tile snapshots rebuilt from the events — expands to exactly rows [a, b) of the run, and its tile index satisfies
    the invariants K1 relies on."""
    blob, inputs, cfg = _CASES[name]
    try:
        log = rt.interpret(blob, inputs, rt.VMConfig(enable_execution_trace=True, **cfg), tile_rows=256)
    except rt.RuntimeError:
        pytest.skip("program errors out")
    n = log.n_rows
    whole = helpers.expand_delta_log(log)
    seg = 300
    cuts, a = [], 0
    while a < n - 1:
        b = min(a + seg, n)
        cuts.append((a, b)); a = b - 1                       # overlap by one row
    cuts += [(1, n), (255, 257), (257, 258), (n - 1, n), (777 % n, min(n, 777 % n + 513))]
    for a, b in cuts:
        sh = log.shard(a, b)
        assert sh.n_rows == b - a and sh.cycle_base == a
        helpers.check_tile_index(sh)
        rows = helpers.expand_delta_log(sh)
        rows["cycle"] += np.uint64(a)
        helpers.assert_rows_equal(rows, whole[a:b])
        sh.close()
    log.close()
