"""BASELINE.json configs[2] and configs[4] at their FULL sizes on one MI355X, through the C ABI.

The oracle cannot hold 2^24 reference rows comfortably (6 GB) and its prover is O(N log N) scalar code, so these tests use what
the task statement prescribes for full sizes: the oracle on SAMPLED row windows (oracle.run(keep_rows=...) executes the whole
program and keeps a window), plus size-independent properties over every row (instruction semantics of the fib loop between
consecutive rows, sortedness of the memory trace, the SHA-256 chain linking every block to the next), plus the oracle's verifier
on the full-size proof."""
import ctypes as C

import numpy as np
import pytest

from oracle import api as oracle, stark_api as so
from zkir_amd import runtime as rt, spec

import helpers

pytestmark = pytest.mark.gpu
M40 = np.uint64((1 << 40) - 1)
P = so.P


def _windows(n, w=384):
    """First rows, a window straddling the middle (and a tile boundary), the last rows."""
    return [(0, w), (n // 2 - w // 2, n // 2 + w // 2), (n // 3 - 7, n // 3 - 7 + w), (n - w, n)]


def test_config2_fib_2p24_trace_commit_proof():
    """configs[2]: 2^24-cycle fib — trace rows vs the oracle on sampled windows, loop semantics on ALL rows, commitment root
    reproduced from sampled Merkle paths + the LDE checked against the trace polynomial, and the full proof accepted by the
    oracle verifier."""
    _fib_full_size(24, _windows(1 << 24))


def test_config3_row_count_2p26_on_one_gpu():
    """The 2^26 rows of configs[3] (there row-sharded over 8 GPUs) on ONE device: 25 GB of trace and 122 GB of field matrices resident
    in HBM; the same checks as at 2^24 (two oracle windows: each costs one 2^26-cycle oracle run)."""
    import torch
    if torch.cuda.mem_get_info()[1] < 250 * (1 << 30):
        pytest.skip("needs ~185 GB of HBM")
    n = 1 << 26
    _fib_full_size(26, [(0, 256), (n - 300, n)])


def _fib_full_size(k, windows):
    import torch
    from zkir_amd import pipeline as pl, stark
    n = 1 << k
    blob = spec.fib_endless_program().to_bytes()
    res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()          # zkir_exec: interpret + H2D + K1
    assert res.cycles == n and res.halt_reason == rt.HaltReason.CycleLimit() and len(res.execution_trace) == n
    tr = res.execution_trace
    for lo, hi in windows:
        want = oracle.run(blob, max_cycles=n, enable_execution_trace=True, keep_rows=(lo, hi))
        assert want.cycles == n
        helpers.assert_rows_equal(tr.rows_window(lo, hi), want.rows)
    # every row: cycle = index, r0 = 0, and the fib loop's semantics between consecutive rows (pc 0x100C: add r4,r1,r2;
    # 0x1010: addi r1,r2,0; 0x1014: addi r2,r4,0; 0x1018: addi r3,r3,-1) — values wrap mod 2^40 (value.rs:592-596)
    cyc = tr.column(rt.FIELD_CYCLE)
    assert np.array_equal(cyc, np.arange(n, dtype=np.uint64))
    del cyc
    pc = tr.column(rt.FIELD_PC)
    regs = {r: tr.column(rt.FIELD_REGISTERS, r) for r in (0, 1, 2, 3, 4)}
    assert not regs[0].any()
    at = lambda a: np.nonzero(pc[:-1] == a)[0]  # noqa: E731
    i = at(0x100C); assert len(i) > n // 7
    assert np.array_equal(regs[4][i + 1], (regs[1][i] + regs[2][i]) & M40)
    i = at(0x1010); assert np.array_equal(regs[1][i + 1], regs[2][i])
    i = at(0x1014); assert np.array_equal(regs[2][i + 1], regs[4][i])
    i = at(0x1018); assert np.array_equal(regs[3][i + 1], (regs[3][i] - np.uint64(1)) & M40)
    for r in (1, 2, 3, 4):                                        # a register only changes at its writer
        writer = {1: 0x1010, 2: 0x1014, 3: 0x1018, 4: 0x100C}[r]
        ch = np.nonzero(regs[r][1:] != regs[r][:-1])[0]
        assert np.isin(pc[ch], [writer, 0x1000, 0x1004, 0x1008]).all()
    bits4 = tr.column(rt.FIELD_BOUND_BITS, 4)                     # the slightly absurd ever-growing bound column: monotone
    assert (np.diff(bits4[16:].astype(np.int64)) >= 0).all() and bits4[-1] > 1_000_000        # (the first write drops it from ProgramWidth(40) to 3)
    del regs, bits4

    # ---- commitment over the same device trace ----
    ctx = stark.StarkContext(k)
    trace_c = tr.columns
    NB = stark.W_MAIN // 8                                                      # B8 layout: blocks of 8 columns, [rows][8]
    m = torch.empty((NB, n, 8), dtype=torch.int32, device="cuda")
    L = torch.empty((NB, 2 * n, 8), dtype=torch.int32, device="cuda")
    tree = torch.empty(4 * (4 * n - 1), dtype=torch.int32, device="cuda")
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = rt.lib()
    pl._check(lib.zkir_main_trace_launch(C.byref(trace_c), n, 0, m.data_ptr(), sp))
    R4 = 21 - 3                                                                  # logical column 21 (r4 limb 0); R0's three limbs are not committed
    cols_m = {c: m[c // 8, :, c % 8].cpu().numpy().view(np.uint32) for c in (0, R4)}     # cycle, r4 limb 0 (LDE clobbers m)
    pl._check(lib.zkir_lde_launch(ctx.handle, m.data_ptr(), stark.W_MAIN, L.data_ptr(), sp))
    pl._check(lib.zkir_merkle_commit_launch(ctx.handle, L.data_ptr(), stark.W_MAIN, 2 * n, tree.data_ptr(), sp))
    root = tree[-4:].cpu().numpy().view(np.uint32)
    assert np.array_equal(cols_m[0], np.arange(n, dtype=np.uint64) % P)           # cycle column of the main trace
    assert np.array_equal(cols_m[R4], (tr.column(rt.FIELD_REGISTERS, 4) & np.uint64(0xFFFFF)).astype(np.uint32))
    from test_gpu_stark import _bary_eval
    rng = np.random.default_rng(k)
    for c in (0, R4):                                             # the LDE is the extension of the trace column: same value at a random point
        z = int(rng.integers(2, P))
        assert _bary_eval(cols_m[c], k, 1, z) == _bary_eval(L[c // 8, :, c % 8].cpu().numpy().view(np.uint32), k + 1, 31, z), c
    for j in [0, 2 * n - 1] + [int(x) for x in rng.integers(0, 2 * n, 4)]:       # sampled leaves: oracle sponge + oracle compression up to the root
        node = so.hash_elems(L[:, j, :].reshape(-1).cpu().numpy().view(np.uint32))
        off, mm = 0, 2 * n
        assert np.array_equal(node, tree[4 * j:4 * j + 4].cpu().numpy().view(np.uint32))
        while mm > 1:
            sib = tree[off + 4 * (j ^ 1): off + 4 * (j ^ 1) + 4].cpu().numpy().view(np.uint32)
            node = so.compress(node, sib) if j % 2 == 0 else so.compress(sib, node)
            off += 4 * mm; mm //= 2; j //= 2
        assert np.array_equal(node, root)
    del m, L, tree
    torch.cuda.empty_cache()

    # ---- full proof of the 2^k-row run, accepted by the oracle verifier; its trace root is the commitment above ----
    pub = res.public_inputs()
    proof = stark.prove(ctx, trace_c, pub)
    assert rt.verify(proof, pub) == 0 and so.verify(proof) == 0                   # the product's verifier and the oracle's
    assert proof[2] == k and proof[7] == n and stark.trace_root(proof) == [int(x) for x in root]
    t = proof.copy(); t[len(t) // 3] = (int(t[len(t) // 3]) + 1) % P
    assert so.verify(t) != 0 and rt.verify(t) == so.verify(t)
    ctx.close(); res.close()


def test_opcode_families_2p22_semantics_on_all_rows_and_proof():
    """AIR v3 at scale: 2^22 cycles of spec.compare_loop_program through zkir_exec — oracle rows on sampled windows, the semantics of
    SUB / SLTU / SGEU / SEQ / SNE and of the four branches checked on EVERY row with numpy (the next row's register / pc), and the full
    proof of the run accepted by both verifiers (every one of its 4 M rows is a constrained class)."""
    from zkir_amd import stark
    k = 22
    n = 1 << k
    blob = spec.compare_loop_program().to_bytes()
    res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()
    assert res.cycles == n and res.halt_reason == rt.HaltReason.CycleLimit()
    tr = res.execution_trace
    for lo, hi in _windows(n, 256):
        want = oracle.run(blob, max_cycles=n, enable_execution_trace=True, keep_rows=(lo, hi))
        helpers.assert_rows_equal(tr.rows_window(lo, hi), want.rows)
    pc = tr.column(rt.FIELD_PC)
    ins = tr.column(rt.FIELD_INSTRUCTION)
    regs = [tr.column(rt.FIELD_REGISTERS, r) for r in range(16)]
    R = np.stack(regs)                                                            # [16][n] (fields b / c of an I-type word are immediate bits: any index)
    op = (ins & 0x7F)[:-1]; fa = ((ins >> 7) & 0xF)[:-1]; fb = ((ins >> 11) & 0xF)[:-1]; fc = ((ins >> 15) & 0xF)[:-1]
    idx = np.arange(n - 1)
    a_, b_, c_ = R[fa, idx], R[fb, idx], R[fc, idx]                               # reg[field a], reg[field b], reg[field c] of every row
    nxt_rd = R[fa, idx + 1]
    imm = ((ins[:-1] >> 15).astype(np.int64) - ((ins[:-1] >> 31).astype(np.int64) << 17))
    seq_pc = pc[:-1] + np.uint64(4); tgt_pc = (pc[:-1].astype(np.int64) + imm).astype(np.uint64)
    checks = {
        0x01: lambda m: np.array_equal(nxt_rd[m], (b_[m] - c_[m]) & M40),
        0x20: lambda m: np.array_equal(nxt_rd[m], ((b_[m] & M40) < (c_[m] & M40)).astype(np.uint64)),
        0x21: lambda m: np.array_equal(nxt_rd[m], ((b_[m] & M40) >= (c_[m] & M40)).astype(np.uint64)),
        0x24: lambda m: np.array_equal(nxt_rd[m], (b_[m] == c_[m]).astype(np.uint64)),
        0x25: lambda m: np.array_equal(nxt_rd[m], (b_[m] != c_[m]).astype(np.uint64)),
        0x40: lambda m: np.array_equal(pc[1:][m], np.where(a_[m] == b_[m], tgt_pc[m], seq_pc[m])),
        0x41: lambda m: np.array_equal(pc[1:][m], np.where(a_[m] != b_[m], tgt_pc[m], seq_pc[m])),
        0x44: lambda m: np.array_equal(pc[1:][m], np.where((a_[m] & M40) < (b_[m] & M40), tgt_pc[m], seq_pc[m])),
        0x45: lambda m: np.array_equal(pc[1:][m], np.where((a_[m] & M40) >= (b_[m] & M40), tgt_pc[m], seq_pc[m])),
    }
    for code, ok in checks.items():
        m = op == code
        assert m.sum() > n // 40 and ok(m), hex(code)
    assert set(int(o) for o in np.unique(op)) == {0x00, 0x01, 0x08, 0x20, 0x21, 0x24, 0x25, 0x40, 0x41, 0x44, 0x45, 0x48}   # nothing runs as class "other" (JAL: the back edge before the counter moves)
    del R, regs, a_, b_, c_, nxt_rd
    ctx = stark.StarkContext(k)
    pub = res.public_inputs()
    proof = stark.prove(ctx, tr.columns, pub)
    assert rt.verify(proof, pub) == 0 and so.verify(proof) == 0
    t = proof.copy(); t[len(t) // 2] = (int(t[len(t) // 2]) + 1) % P
    assert so.verify(t) != 0 and rt.verify(t) == so.verify(t)
    ctx.close(); res.close()


def test_call_loop_2p20_control_flow_on_all_rows_and_proof():
    """AIR v4 at scale: 2^20 cycles of spec.call_loop_program — every JALR lands on (rs1 + imm) & ~1 and links pc + 4, every row that is not a
    branch or a jump advances the pc by 4 (checked on ALL rows with numpy), and the proof of the run is accepted by both verifiers."""
    from zkir_amd import stark
    k = 20
    n = 1 << k
    blob = spec.call_loop_program().to_bytes()
    res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()
    assert res.cycles == n
    tr = res.execution_trace
    lo, hi = n // 2 - 100, n // 2 + 100
    helpers.assert_rows_equal(tr.rows_window(lo, hi), oracle.run(blob, max_cycles=n, enable_execution_trace=True, keep_rows=(lo, hi)).rows)
    pc = tr.column(rt.FIELD_PC); ins = tr.column(rt.FIELD_INSTRUCTION)
    R = np.stack([tr.column(rt.FIELD_REGISTERS, r) for r in range(16)])
    op = (ins & 0x7F)[:-1]; fa = ((ins >> 7) & 0xF)[:-1]; fb = ((ins >> 11) & 0xF)[:-1]
    idx = np.arange(n - 1)
    imm = ((ins[:-1] >> 15).astype(np.int64) - ((ins[:-1] >> 31).astype(np.int64) << 17))
    j = op == 0x49
    assert j.sum() > n // 30
    assert np.array_equal(pc[1:][j], (R[fb, idx][j].astype(np.int64) + imm[j]).astype(np.uint64) & ~np.uint64(1))
    w = j & (fa != 0)
    assert np.array_equal(R[fa, idx + 1][w], pc[:-1][w] + np.uint64(4))
    seq = ~np.isin(op, [0x40, 0x41, 0x42, 0x43, 0x44, 0x45, 0x48, 0x49])
    assert seq.sum() > n // 3 and np.array_equal(pc[1:][seq], pc[:-1][seq] + np.uint64(4))
    del R
    ctx = stark.StarkContext(k)
    pub = res.public_inputs()
    proof = stark.prove(ctx, tr.columns, pub)
    assert rt.verify(proof, pub) == 0 and so.verify(proof) == 0 and proof[3] == stark.W_MAIN
    ctx.close(); res.close()


def test_signed_comparisons_and_conditional_moves_2p20_semantics_on_all_rows_and_proof():
    """AIR v5 / v6 at scale: 2^20 cycles of spec.signed_loop_program and 2^18 of spec.cmov_loop_program through zkir_exec — the semantics of SLT / SGE /
    BLT / BGE (Value40::signed_lt, value.rs:710-716) and of CMOV / CMOVZ / CMOVNZ (execute.rs:434-472) checked on EVERY row with numpy against the
    next row's register / pc, oracle rows on a window, and the proofs of both runs accepted by both verifiers."""
    from zkir_amd import stark
    M40 = np.uint64((1 << 40) - 1); SB = np.uint64(1 << 39)
    for prog, k in ((spec.signed_loop_program, 20), (spec.cmov_loop_program, 18)):
        n = 1 << k
        blob = prog().to_bytes()
        res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()
        assert res.cycles == n
        tr = res.execution_trace
        lo, hi = n // 2 - 100, n // 2 + 100
        helpers.assert_rows_equal(tr.rows_window(lo, hi), oracle.run(blob, max_cycles=n, enable_execution_trace=True, keep_rows=(lo, hi)).rows)
        pc = tr.column(rt.FIELD_PC); ins = tr.column(rt.FIELD_INSTRUCTION)
        R = np.stack([tr.column(rt.FIELD_REGISTERS, r) for r in range(16)])
        op = (ins & 0x7F)[:-1]; fa = ((ins >> 7) & 0xF)[:-1]; fb = ((ins >> 11) & 0xF)[:-1]; fc = ((ins >> 15) & 0xF)[:-1]
        idx = np.arange(n - 1)
        a_, b_, c_ = R[fa, idx], R[fb, idx], R[fc, idx]
        nxt_rd = R[fa, idx + 1]
        imm = ((ins[:-1] >> 15).astype(np.int64) - ((ins[:-1] >> 31).astype(np.int64) << 17))
        seq_pc = pc[:-1] + np.uint64(4); tgt_pc = (pc[:-1].astype(np.int64) + imm).astype(np.uint64)
        slt = lambda x, y: ((x & M40) ^ SB) < ((y & M40) ^ SB)                           # noqa: E731
        keep = lambda m: (R[:, idx + 1][:, m] == R[:, idx][:, m]).all()                  # noqa: E731   nothing at all changes
        checks = {
            0x22: lambda m: np.array_equal(nxt_rd[m], slt(b_[m], c_[m]).astype(np.uint64)),
            0x23: lambda m: np.array_equal(nxt_rd[m], (~slt(b_[m], c_[m])).astype(np.uint64)),
            0x42: lambda m: np.array_equal(pc[1:][m], np.where(slt(a_[m], b_[m]), tgt_pc[m], seq_pc[m])),
            0x43: lambda m: np.array_equal(pc[1:][m], np.where(~slt(a_[m], b_[m]), tgt_pc[m], seq_pc[m])),
            0x26: lambda m: np.array_equal(nxt_rd[m & (c_ != 0) & (fa != 0)], b_[m & (c_ != 0) & (fa != 0)]) and keep(m & ((c_ == 0) | (fa == 0))),
            0x28: lambda m: np.array_equal(nxt_rd[m & (c_ != 0) & (fa != 0)], b_[m & (c_ != 0) & (fa != 0)]) and keep(m & ((c_ == 0) | (fa == 0))),
            0x27: lambda m: np.array_equal(nxt_rd[m & (c_ == 0) & (fa != 0)], b_[m & (c_ == 0) & (fa != 0)]) and keep(m & ((c_ != 0) | (fa == 0))),
        }
        seen = 0
        for code, ok in checks.items():
            m = op == code
            if m.any():
                assert m.sum() > n // 64 and ok(m), hex(code)
                seen += 1
        assert seen == (4 if prog is spec.signed_loop_program else 3)
        del R, a_, b_, c_, nxt_rd
        ctx = stark.StarkContext(k)
        pub = res.public_inputs()
        proof = stark.prove(ctx, tr.columns, pub)
        assert rt.verify(proof, pub) == 0 and so.verify(proof) == 0 and proof[3] == stark.W_MAIN
        ctx.close(); res.close()


def test_config4_sha_chain_2p22_syscall_chip_columns():
    """configs[4]: SHA-256 hash-chain program for 2^22 cycles — trace rows, memory ops (row order, CSR, sorted), and the SHA-256
    chip columns, all behind the drop-in handle (zkir_result_*), vs the oracle on sampled windows / blocks and through full-size
    properties."""
    import hashlib
    k = 22
    n = 1 << k
    blob = spec.sha256_chain_program().to_bytes()
    res = rt.VM(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)).run()
    assert res.cycles == n and len(res.execution_trace) == n
    w = res.memory_witness()
    n_ops = w.n_ops
    assert n_ops == res.memory_op_count() and n_ops > 27_000_000
    offs = res._d2h(w.row_offsets, n + 1, "<u8")
    assert offs[0] == 0 and offs[-1] == n_ops and (np.diff(offs.astype(np.int64)) >= 0).all()

    def ops_window(cols, a, b):
        out = np.zeros(b - a, dtype=rt._MEMOP_DTYPE)
        for name in rt._MEMOP_DTYPE.names:
            elt = rt._MEMOP_DTYPE[name].itemsize
            out[name] = res._d2h(getattr(cols, name) + a * elt, b - a, rt._MEMOP_DTYPE[name])
        return out

    for lo, hi in _windows(n):
        want = oracle.run(blob, max_cycles=n, enable_execution_trace=True, keep_rows=(lo, hi))
        helpers.assert_rows_equal(res.execution_trace.rows_window(lo, hi), want.rows)
        a, b = int(offs[lo]), int(offs[hi])
        assert b - a == len(want.memops)
        assert np.array_equal(offs[lo:hi + 1] - offs[lo], want.row_memop_offsets)
        assert np.array_equal(ops_window(w.row_order, a, b), want.memops)
        assert np.array_equal(ops_window(w.sorted, a, b), want.sorted_memops)      # get_memory_trace sorts by timestamp first: windows are self-contained
    # full size: the sorted trace is sorted by (timestamp, address, Read<Write) and is a permutation of the row-order ops
    ts = res._d2h(w.sorted.timestamp, n_ops, "<u8"); ad = res._d2h(w.sorted.address, n_ops, "<u8"); wr = res._d2h(w.sorted.is_write, n_ops, "u1")
    same_t = ts[1:] == ts[:-1]
    assert (ts[1:] >= ts[:-1]).all()
    assert (ad[1:][same_t] >= ad[:-1][same_t]).all()
    same_ta = same_t & (ad[1:] == ad[:-1])
    assert (wr[1:][same_ta] >= wr[:-1][same_ta]).all()
    ad0 = res._d2h(w.row_order.address, n_ops, "<u8"); val0 = res._d2h(w.row_order.value, n_ops, "<u8"); val = res._d2h(w.sorted.value, n_ops, "<u8")
    mix = lambda a, v: int((a * np.uint64(0x9E3779B97F4A7C15) ^ v).sum(dtype=np.uint64))  # noqa: E731  order-independent checksum
    assert mix(ad, val) == mix(ad0, val0)
    del ts, ad, wr, ad0, val0, val

    # ---- SHA-256 chip: 608 word-columns per single-block call ----
    cols, stamps = res.sha256_witnesses()
    nb = cols.shape[1]
    assert nb == len(res.delta_log.sha_blocks) and nb > 690_000
    assert (np.diff(stamps.astype(np.int64)) > 0).all()
    # full size: it is a hash CHAIN — the digest of block b is the message of block b+1.  The syscall stores the eight big-endian
    # digest words with little-endian write_u32 (crypto.rs:252-255), so in memory every word is byte-swapped; padding words are fixed
    assert np.array_equal(cols[600:608, :-1].byteswap(), cols[0:8, 1:])
    assert (cols[8] == 0x80000000).all() and not cols[9:15].any() and (cols[15] == 256).all()
    assert np.array_equal(cols[16:24], np.repeat(np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19],
                                                           dtype=np.uint32)[:, None], nb, axis=1))
    assert np.array_equal(cols[24:40], cols[0:16])                # W[0..16) = the message block
    assert np.array_equal(cols[600:608], cols[16:24] + cols[88 + 8 * 63: 88 + 8 * 64])   # final = H0 + last round state (mod 2^32)
    msg = bytes(range(32))
    for b in list(range(4)) + [nb // 2, nb - 1]:                  # sampled blocks: every word vs the oracle's witness, digest vs hashlib
        blk = cols[0:8, b].astype(">u4").tobytes()
        if b < 4:
            assert blk == msg
            dg = hashlib.sha256(msg).digest()
            msg = b"".join(dg[4 * i:4 * i + 4][::-1] for i in range(8))       # what the syscall left in memory
        wit = oracle.sha256_witness(blk, int(stamps[b]))
        assert np.array_equal(cols[:, b], wit["flat"])
        assert cols[600:608, b].astype(">u4").tobytes() == hashlib.sha256(blk).digest()
    res.close()


@pytest.mark.parametrize("log_n", [24, 25, 26])
def test_lde_matches_oracle_at_config_sizes(log_n):
    """The LDE pass structures of the big configs against the oracle DIRECTLY: 14 / 15 / 16 strided stages on each side of the fused
    middle = 10 + 4 (2^24), 10 + 3 + 2 (2^25), 10 + 6 (2^26) (ntt.hip: run_strided_stages).  One B8 block, two random columns (the oracle's textbook NTT
    takes ~1 minute per 2^26-point column), the other six zero — they must come out zero."""
    import torch
    from zkir_amd import stark
    if torch.cuda.mem_get_info()[1] < (28 << log_n) + (4 << 30):
        pytest.skip("not enough HBM")
    n = 1 << log_n
    rng = np.random.default_rng(1000 + log_n)
    mat = rng.integers(0, P, (2, n)).astype(np.uint32)
    ctx = stark.StarkContext(log_n)
    out = stark.lde(ctx, stark.to_b8(torch.from_numpy(mat.view(np.int32)).cuda()), clobber=True)
    assert not out[0, :, 2:].any()
    got = stark.from_b8(out, 2).cpu().numpy().view(np.uint32)
    del out
    for k in range(2):
        assert np.array_equal(got[k], so.lde(mat[k], 1)[1]), f"column {k}"
    ctx.close()


@pytest.mark.parametrize("k", [16, 18, 20, 22, 24])
def test_commitment_root_equals_the_oracles_at_config_size(k):
    """BASELINE configs[1] says "bit-exact root vs CPU": the GPU's trace-commitment root of the 2^20-cycle fib run (and of two smaller sizes) equals the root
    the CPU oracle computed for it — tests/golden/config_roots.json, written by tests/golden/make_config_roots.py from the oracle alone (4 minutes of
    textbook arithmetic at 2^20: too slow to repeat in every test run, which is why it is a fixture).  k = 24 is BASELINE configs[2]'s own size (round 5: the
    oracle's column-blocked commit, so_commit_trace_blocked — eight columns at a time, one sponge state per leaf; held equal to the plain one by tests/test_stark_oracle.py)."""
    import json
    import os
    from zkir_amd import pipeline as pl, stark
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_roots.json")))
    if str(k) not in gold["roots"]:
        pytest.skip(f"no oracle root for 2^{k} in the fixture")
    blob = spec.fib_endless_program().to_bytes()
    assert blob.hex() == gold["program_blob_hex"]
    n = 1 << k
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
    ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
    ctx = stark.StarkContext(k)
    root, _, _ = stark.commit_trace(ctx, tr)
    assert [int(x) for x in root] == gold["roots"][str(k)]["root"]
    # the proof of the same run commits to the same trace
    proof = stark.prove(ctx, tr, rt.public_inputs(log, blob))
    assert stark.trace_root(proof) == gold["roots"][str(k)]["root"]
    ctx.close(); log.close()


@pytest.mark.parametrize("k", [16, 22])
def test_config4_sha_chain_is_provable_in_mode4(k):
    """BASELINE configs[4]'s program — the SHA-256 hash chain — PROVEN (VERDICT r5 missing #4: "witnessed, not provable"): AIR mode 4 carries one record per hash syscall and the
    verifier recomputes every digest (oracle/stark_oracle.cpp "MODE 4 (b)").  2^16 cycles: the GPU proof equals the oracle's word for word.  2^22 cycles (the config's size:
    699 k SHA-256 calls, 1.05 M touched-cell records): the GPU proof is accepted by BOTH verifiers — each of which hashes the 699 k messages itself — and a proof with one
    message bit flipped in the tape is rejected by both."""
    from zkir_amd import pipeline as pl, stark
    blob = spec.sha256_chain_program().to_bytes()
    n = 1 << k
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
    ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
    pub = rt.public_inputs(log, blob, [], wide_mode=True, mem_witness="host")
    n_calls = pub._mem_ref.n_hash_calls
    assert n // 6 - 16 <= n_calls <= n // 6                                      # one hash syscall per six-row iteration, after the copy loop that sets the seed up
    ctx = stark.StarkContext(k)
    proof = stark.prove(ctx, tr, pub)
    assert proof[1] == 12 and proof[9] == 4 and proof[3] == 288
    opub = so.public_inputs(n, blob, [], [], (2, 0), wide_mode=True)
    assert rt.verify(proof, pub) == 0 and so.verify(proof, opub) == 0
    lay = stark.proof_layout(proof)
    at = lay["hash_section"]
    assert int(proof[at]) == n_calls
    t = proof.copy(); t[at + 1 + 8 + 1] ^= 1                                   # one bit of the first call's message (its first cell's bytes)
    assert rt.verify(t) == so.verify(t) != 0                                   # (the tape is in the transcript: the first check that breaks is the grinding nonce's, 12; a prover who re-grinds meets 10 — tests/test_stark_mode4.py)
    if k == 16:
        ores = oracle.run(blob, max_cycles=n, enable_execution_trace=True)
        assert np.array_equal(proof, so.prove(ores.rows, opub))
    ctx.close(); log.close()


def test_config3_row_sharded_commitment_equals_the_oracles_at_full_size():
    """BASELINE configs[3]'s OWN workload on one device: the 2^26-cycle fib run cut into 8 row shards of 2^23 rows; every shard goes through what one rank
    of `bench.py --gpus 8` runs — zkir_interpret_window (rows before the shard executed untraced, the shard traced), upload, trace fill, main trace, LDE, Merkle subtree — one
    after the other on this GPU, and the 8 subtree roots are capped by zkir_merkle_cap_launch (what every rank computes after the all-gather).  Every subtree root and the
    capped root equal tests/golden/config_roots.json["26x8"], which the CPU oracle computed shard by shard (make_config_roots.py sharded 26 8: 47 minutes on 6 threads).
    With it configs[3]'s data path is bit-exact at full size everywhere except the xGMI hop itself (16 bytes per rank)."""
    import json
    import os
    import torch
    from zkir_amd import pipeline as pl, stark
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_roots.json")))
    g = gold["roots"].get("26x8")
    if not g or "root" not in g:
        pytest.skip("no oracle answer for the 8 x 2^23 sharded run in the fixture")
    blob = spec.fib_endless_program().to_bytes()
    assert blob.hex() == gold["program_blob_hex"]
    G, n, total = g["shards"], g["rows_per_shard"], g["rows"]
    ctx = stark.StarkContext(23)
    roots = []
    for r in range(G):
        log = rt.interpret(blob, [], rt.VMConfig(max_cycles=total, enable_execution_trace=True), window=(r * n, (r + 1) * n))
        assert log.n_rows == n and log.cycle_base == r * n
        ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
        root, L, tree = stark.commit_trace(ctx, tr)
        assert [int(x) for x in root] == g["shard_roots"][r], f"shard {r}"
        roots.append(tree[-4:].clone())
        del L, tree, tr, ddl
        log.close()
        torch.cuda.empty_cache()
    top = stark.merkle_cap(ctx, torch.stack(roots)).cpu().numpy().view(np.uint32)
    assert [int(x) for x in top] == g["root"]
    ctx.close()


@pytest.mark.parametrize("k", [12, 16, 18, 20, 22, 23, 24])
def test_proof_equals_the_oracles_at_config_size(k):
    """BASELINE's metric is "end-to-end prove ms, 2^20-cycle fib" and north_star asks for bit-identical proof bytes: the GPU prover's COMPLETE proof of that run
    (and of three smaller sizes) equals the proof the CPU oracle computed for it — tests/golden/config_proofs.json (length, SHA-256 of the words, 257 spaced
    words), written by tests/golden/make_config_proofs.py from the oracle alone (~13 minutes of textbook arithmetic at 2^20: a fixture, not a per-run
    computation).  tests/test_gpu_stark.py compares whole proofs word for word up to 2^13 rows; this is the same statement at the headline size.
    k = 24 is BASELINE configs[2] as worded ("2^24-cycle fib ... end-to-end proof bit-exact"; round 6): the golden comes from the oracle's memory-lean threaded prover
    so::prove_lean, which tests/test_stark_oracle.py::test_lean_prover_equals_the_plain_one holds equal to so::prove word for word in every mode."""
    import hashlib
    import json
    import os
    from zkir_amd import pipeline as pl, stark
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_proofs.json")))
    blob = spec.fib_endless_program().to_bytes()
    assert blob.hex() == gold["program_blob_hex"]
    if str(k) not in gold["proofs"]:
        pytest.skip(f"no oracle proof for 2^{k} in the fixture")
    g = gold["proofs"][str(k)]
    n = 1 << k
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True))
    ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
    ctx = stark.StarkContext(k)
    pub = rt.public_inputs(log, blob)
    proof = np.ascontiguousarray(stark.prove(ctx, tr, pub), dtype="<u4")
    assert len(proof) == g["words"]
    pos = [int(i * (len(proof) - 1) // (len(g["samples"]) - 1)) for i in range(len(g["samples"]))]
    bad = [p for p, w in zip(pos, g["samples"]) if int(proof[p]) != w]
    assert not bad, f"proof differs from the oracle's at sampled words {bad[:8]} (of {len(proof)})"
    assert hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
    assert rt.verify(proof, pub) == 0
    ctx.close(); log.close()


@pytest.mark.parametrize("k", [12, 16, 20])
def test_proof_at_the_second_parameter_set_equals_the_oracles(k):
    """zkir_prover_params (round 5): the fib run proven with 84 FRI queries and 16 grinding bits — the GPU prover's complete proof equals the one the CPU oracle computed
    with the same parameters (tests/golden/config_proofs.json: fib_84q_16b_proofs), says so in header words 4 and 6, and a verifier expecting the defaults refuses it."""
    import hashlib
    import json
    import os
    from zkir_amd import pipeline as pl, stark
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_proofs.json")))
    if str(k) not in gold.get("fib_84q_16b_proofs", {}):
        pytest.skip(f"no oracle proof at the second parameter set for 2^{k} in the fixture")
    g = gold["fib_84q_16b_proofs"][str(k)]
    blob = spec.fib_endless_program().to_bytes()
    assert blob.hex() == gold["program_blob_hex"]
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=1 << k, enable_execution_trace=True))
    ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
    ctx = stark.StarkContext(k)
    pub = rt.public_inputs(log, blob, num_queries=g["num_queries"], pow_bits=g["pow_bits"])
    proof = np.ascontiguousarray(stark.prove(ctx, tr, pub), dtype="<u4")
    assert (int(proof[4]), int(proof[6])) == (84, 16) and len(proof) == g["words"]
    pos = [int(i * (len(proof) - 1) // (len(g["samples"]) - 1)) for i in range(len(g["samples"]))]
    bad = [p for p, w in zip(pos, g["samples"]) if int(proof[p]) != w]
    assert not bad, f"proof differs from the oracle's at sampled words {bad[:8]} (of {len(proof)})"
    assert hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
    assert rt.verify(proof, pub) == 0 and rt.verify(proof) == 0 and rt.verify(proof, rt.public_inputs(log, blob)) == 2
    ctx.close(); log.close()


@pytest.mark.parametrize("k", [12, 16, 20])
def test_mode2_proof_equals_the_oracles_and_says_what_was_output(k):
    """MODE 2 (the I/O argument) at the headline size (VERDICT r4 task 2): a fib run that halts by itself a few rows short of 2^k and WRITES fib(cnt + 1) mod 2^40 —
    the GPU prover's complete mode-2 proof equals the one the CPU oracle computed (tests/golden/config_proofs.json: mode2_fib_out_proofs, written by
    make_config_proofs.py mode2 with its own encoder), and the output tape the proof carries is the run's output."""
    import hashlib
    import json
    import os
    from zkir_amd import pipeline as pl, stark
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_proofs.json")))
    if str(k) not in gold.get("mode2_fib_out_proofs", {}):
        pytest.skip(f"no mode-2 oracle proof for 2^{k} in the fixture")
    g = gold["mode2_fib_out_proofs"][str(k)]
    blob = bytes.fromhex(g["program_blob_hex"])
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=2 << k, enable_execution_trace=True))
    assert log.n_rows == g["rows"] and list(log.outputs) == [g["output"]] and log.halt_reason == rt.HaltReason.Exit(0)
    ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
    ctx = stark.StarkContext(k)
    pub = rt.public_inputs(log, blob, [], io_mode=True)
    proof = np.ascontiguousarray(stark.prove(ctx, tr, pub), dtype="<u4")
    assert proof[9] == 2 and len(proof) == g["words"]
    pos = [int(i * (len(proof) - 1) // (len(g["samples"]) - 1)) for i in range(len(g["samples"]))]
    bad = [p for p, w in zip(pos, g["samples"]) if int(proof[p]) != w]
    assert not bad, f"mode-2 proof differs from the oracle's at sampled words {bad[:8]} (of {len(proof)})"
    assert hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
    assert rt.verify(proof, pub) == 0
    assert rt.verify_io(proof, pub, [], [g["output"]], rt.HaltReason.Exit(0)) == 0
    assert rt.verify_io(proof, pub, [], [g["output"] ^ 1], rt.HaltReason.Exit(0)) != 0          # .. and not any other output
    ctx.close(); log.close()


@pytest.mark.parametrize("k", [14, 16, 18])
def test_mode3_proof_equals_the_oracles_on_the_memory_ring(k):
    """MODE 3 (memory argument, bitwise opcodes, shifts, MUL) beyond the small programs of tests/test_gpu_stark.py: the GPU prover's complete mode-3 proof of the
    memory-ring walk halted at 2^k cycles (1024 cells, each re-visited 2^(k-14) times over: 5 memory accesses and 3 bitwise opcodes in every 16 rows; the memory witness made
    on the device by the sort + segmented scan) equals the proof the CPU oracle computed for it (tests/golden/config_proofs.json: mode3_ring_proofs)."""
    import hashlib
    import json
    import os
    from zkir_amd import pipeline as pl, stark
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_proofs.json")))
    blob = spec.memory_ring_program(10).to_bytes()
    assert blob.hex() == gold["ring_program_blob_hex"]
    g = gold["mode3_ring_proofs"][str(k)]
    log = rt.interpret(blob, [], rt.VMConfig(max_cycles=1 << k, enable_execution_trace=True))
    ddl = pl.upload(log); tr = pl.DeviceTrace(ddl); pl.trace_fill(pl.trace_fill_args(ddl, tr))
    ctx = stark.StarkContext(k)
    pub = rt.public_inputs(log, blob, [], mem_mode=True, mem_witness="device")
    proof = np.ascontiguousarray(stark.prove(ctx, tr, pub), dtype="<u4")
    assert proof[9] == 3 and len(proof) == g["words"]
    pos = [int(i * (len(proof) - 1) // (len(g["samples"]) - 1)) for i in range(len(g["samples"]))]
    bad = [p for p, w in zip(pos, g["samples"]) if int(proof[p]) != w]
    assert not bad, f"mode-3 proof differs from the oracle's at sampled words {bad[:8]} (of {len(proof)})"
    assert hashlib.sha256(proof.tobytes()).hexdigest() == g["sha256"]
    assert rt.verify(proof, pub) == 0
    ctx.close(); log.close()
