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


def _same_log(a, b, what):
    """Two delta logs describe the same rows: every array equal, same base / tiling."""
    assert a.n_rows == b.n_rows and a.cycle_base == b.cycle_base and a.tile_rows == b.tile_rows, what
    for name in ("pc", "inst", "reg_events", "tile_ev_off", "tile_snap", "mem_events", "rc_events", "rc_offsets", "rc_cycles", "norm_events", "sha_blocks"):
        x, y = getattr(a, name), getattr(b, name)
        assert x.shape == y.shape and np.array_equal(x, y), f"{what}: {name} differs"


@pytest.mark.parametrize("name", sorted(_CASES))
def test_trace_windows_equal_the_shards_of_the_whole_run(name):
    """zkir_interpret_window (multi-GPU: rank g fast-forwards untraced to its first row, then traces its own rows) records exactly what
    cutting rows [a, b) out of the whole run's log gives — tile-aligned and unaligned windows, windows that reach past the end of
    the run, empty ones — and a window can itself be sharded with absolute row numbers (commit shard + overlapping segment shard)."""
    blob, inputs, cfg = _CASES[name]
    vmc = rt.VMConfig(enable_execution_trace=True, **cfg)
    try:
        log = rt.interpret(blob, inputs, vmc, tile_rows=256)
    except rt.RuntimeError:
        pytest.skip("program errors out")
    n = log.n_rows
    whole = helpers.expand_delta_log(log)
    for a, b in [(0, n), (256, 768), (0, 256), (512, n), (512, n + 1000), (300, 811), (n - 1, n), (1, 2), (n, n + 5), (n + 7, n + 9), (255, 257)]:
        if a > b:
            continue
        win = rt.interpret(blob, inputs, vmc, tile_rows=256, window=(a, b))
        lo, hi = min(a, n), min(b, n)
        assert win.n_rows == hi - lo and win.cycle_base == lo, (a, b)
        assert win.window_open == (b < n), (a, b)                 # stopped at the window's end, or at the run's halt
        if hi > lo:
            helpers.check_tile_index(win)
            rows = helpers.expand_delta_log(win)
            rows["cycle"] += np.uint64(lo)
            helpers.assert_rows_equal(rows, whole[lo:hi])
            ref = log.shard(lo, hi)
            if lo % 256 == 0:                                     # same tiling: the logs are identical array by array
                _same_log(win, ref, f"window {a}:{b}")
            else:
                assert np.array_equal(win.mem_events, ref.mem_events) and np.array_equal(win.rc_events, ref.rc_events) and np.array_equal(win.rc_cycles, ref.rc_cycles)
                assert np.array_equal(win.norm_events, ref.norm_events) and np.array_equal(win.sha_blocks, ref.sha_blocks)
            ref.close()
        if not win.window_open:
            assert win.cycles == log.cycles and win.halt_reason == log.halt_reason and win.outputs == log.outputs
        else:
            assert win.cycles == b
        win.close()
    # a window sharded further, absolute rows: what bench.py --gpus N does (commit shard [g n, (g+1) n) and segment shard [g (n-1), g (n-1) + n))
    a, b = (300, min(n, 1100)) if n > 900 else (n // 4, n - 3)
    win = rt.interpret(blob, inputs, vmc, tile_rows=256, window=(a, b))
    for lo, hi in [(a, b), (a + 1, b), (a + 212, min(b, a + 212 + 256)), (min(512, b - 1), b), (a, a + 1)]:
        sh, ref = win.shard(lo, hi), log.shard(lo, hi)
        rows = helpers.expand_delta_log(sh)
        rows["cycle"] += np.uint64(lo)
        helpers.assert_rows_equal(rows, whole[lo:hi])
        assert sh.cycle_base == lo and np.array_equal(sh.mem_events, ref.mem_events) and np.array_equal(sh.norm_events, ref.norm_events)
        assert np.array_equal(sh.sha_blocks, ref.sha_blocks) and np.array_equal(sh.rc_events, ref.rc_events) and np.array_equal(sh.rc_cycles, ref.rc_cycles)
        sh.close(); ref.close()
    with pytest.raises(rt.RuntimeError):
        win.shard(a - 1, b)
    win.close(); log.close()
