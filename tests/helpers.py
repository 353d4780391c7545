"""Test-only helpers: a numpy expansion of the product's delta log (so the host interpreter can be
checked against the oracle without a GPU) and record-wise comparison utilities."""
from __future__ import annotations

import numpy as np

from oracle import api as oracle
from zkir_amd import runtime as rt

ROW_DTYPE = oracle.ROW_DTYPE


def expand_delta_log(log: rt.DeltaLog) -> np.ndarray:
    """CPU restatement of what K1 (trace_fill.hip) must produce from a delta log.  tests only."""
    n = log.n_rows
    rows = np.zeros(n, dtype=ROW_DTYPE)
    rows["cycle"] = np.arange(n, dtype=np.uint64)
    rows["pc"] = log.pc
    rows["instruction"] = log.inst
    ev = log.reg_events
    for r in range(16):
        e = ev[ev["reg"] == r]
        # event k is visible at rows >= vis[k]; row i takes the last event with vis <= i
        pos = np.searchsorted(e["vis"], np.arange(n, dtype=np.uint64), side="right") - 1
        assert (pos >= 0).all()
        rows["registers"][:, r] = e["value"][pos]
        rows["bound_bits"][:, r] = e["max_bits"][pos]
        rows["bound_tag"][:, r] = e["tag"][pos]
        rows["bound_payload"][:, r] = e["payload"][pos]
        rows["reg_state"][:, r] = e["state"][pos]
    return rows


def check_tile_index(log: rt.DeltaLog):
    """tile_ev_off / tile_snap must describe the event log exactly (they are what K1 trusts)."""
    ev = log.reg_events
    T = log.tile_rows
    n_tiles = (log.n_rows + T - 1) // T
    assert log.n_tiles == n_tiles
    if log.n_rows == 0:
        return
    assert (np.diff(ev["vis"].astype(np.int64)) >= 0).all(), "events must be ordered by vis"
    assert list(ev["reg"][:16]) == list(range(16)) and (ev["vis"][:16] == 0).all()
    # at most one event per (reg, vis)
    key = ev["vis"].astype(np.uint64) * 16 + ev["reg"]
    assert len(np.unique(key)) == len(key)
    for t in range(n_tiles):
        row0 = t * T
        lo = int(np.searchsorted(ev["vis"], row0, side="right"))
        assert log.tile_ev_off[t] == lo, (t, log.tile_ev_off[t], lo)
        for r in range(16):
            idx = np.nonzero(ev["reg"][:lo] == r)[0]
            assert log.tile_snap[t, r] == idx[-1]
    assert log.tile_ev_off[n_tiles] == len(ev)


def assert_rows_equal(got: np.ndarray, want: np.ndarray):
    assert got.shape == want.shape, (got.shape, want.shape)
    for name in ROW_DTYPE.names:
        if not np.array_equal(got[name], want[name]):
            bad = np.nonzero(got[name] != want[name])
            i = bad[0][0]
            raise AssertionError(f"column {name} differs first at row {i}: got {got[name][i]} want {want[name][i]}")


def memops_from_log(log: rt.DeltaLog) -> np.ndarray:
    """Reference-shaped MemoryOp records (row order) from the compact mem events (bound = TypeWidth(8*width))."""
    me = log.mem_events
    out = np.zeros(len(me), dtype=oracle.MEMOP_DTYPE)
    out["address"] = me["address"]; out["value"] = me["value"]; out["timestamp"] = me["row"]
    out["is_write"] = me["is_write"]; out["width"] = me["width"]
    out["bound_bits"] = me["width"].astype(np.uint32) * 8
    out["bound_tag"] = 1
    out["bound_payload"] = me["width"].astype(np.uint64) * 8
    return out


def rc_from_log(log: rt.DeltaLog) -> np.ndarray:
    """RangeCheckWitness entries from (value, pc): chunk decomposition of range_check.rs:175-192."""
    ev = log.rc_events
    out = np.zeros(len(ev), dtype=oracle.RC_DTYPE)
    out["value"] = ev["value"]; out["pc"] = ev["pc"]
    cb = log.rc_chunk_bits
    mask = (1 << cb) - 1
    for l in range(2):
        limb = (ev["value"] >> np.uint64(20 * l)) & np.uint64(0xFFFFF)
        out["chunks"][:, 2 * l] = limb & np.uint64(mask)
        out["chunks"][:, 2 * l + 1] = (limb >> np.uint64(cb)) & np.uint64(mask)
    return out


def norm_from_log(log: rt.DeltaLog) -> np.ndarray:
    """NormalizationEvent records from the compact (raw_value, state) form (normalize.rs:133-153)."""
    ev = log.norm_events
    out = np.zeros(len(ev), dtype=oracle.NORM_DTYPE)
    out["cycle"] = ev["cycle"]; out["pc"] = ev["pc"]; out["reg"] = ev["reg"]; out["opcode"] = ev["opcode"]
    out["normalized_bits"] = 20; out["limb_bits"] = 30; out["cause"] = 0
    bits = np.where(ev["state"] == 0, 20, 30).astype(np.uint64)
    mask = (np.uint64(1) << bits) - np.uint64(1)
    a0 = ev["raw_value"] & mask
    a1 = (ev["raw_value"] >> bits) & mask
    c0 = a0 >> np.uint64(20)
    n0 = a0 & np.uint64(0xFFFFF)
    t = a1 + c0
    out["accumulated"][:, 0] = a0; out["accumulated"][:, 1] = a1
    out["normalized"][:, 0] = n0; out["normalized"][:, 1] = t & np.uint64(0xFFFFF)
    out["carries"][:, 0] = c0; out["carries"][:, 1] = t >> np.uint64(20)
    return out
