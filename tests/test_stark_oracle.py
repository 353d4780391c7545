"""Algebraic self-checks of the self-defined prover-stage oracle (oracle/stark_oracle.cpp).  These stages have no
reference implementation (parity unpinned), so the oracle is pinned by independent python-int
arithmetic here: field laws, naive DFT vs NTT, inverse round trips, LDE agreement on the original domain, Poseidon2
structural properties, Merkle recomputation."""
import numpy as np
import pytest

from bench_cpu import api as cpu_port
from oracle import api as oracle, stark_api as so
from zkir_amd import spec

P = so.P


def test_field_constants_and_roots():
    assert so.lib().so_p() == P == 2**31 - 2**27 + 1
    w27 = so.lib().so_root_of_unity(27)
    assert w27 == pow(31, 15, P) == 0x1A427A41
    assert pow(w27, 1 << 27, P) == 1 and pow(w27, 1 << 26, P) == P - 1
    for k in (1, 2, 10, 20, 21):
        w = so.lib().so_root_of_unity(k)
        assert pow(w, 1 << k, P) == 1 and pow(w, 1 << (k - 1), P) == P - 1
    rng = np.random.default_rng(0)
    for a, b in rng.integers(0, P, (50, 2)):
        assert so.lib().so_fmul(int(a), int(b)) == int(a) * int(b) % P
    assert so.lib().so_fmul(12345, so.lib().so_finv(12345)) == 1


def _emul_ref(a, b):
    t = [0] * 7
    for i in range(4):
        for j in range(4):
            t[i + j] = (t[i + j] + int(a[i]) * int(b[j])) % P
    return [(t[0] + 11 * t[4]) % P, (t[1] + 11 * t[5]) % P, (t[2] + 11 * t[6]) % P, t[3]]


def test_extension_field():
    rng = np.random.default_rng(1)
    for _ in range(20):
        a, b = rng.integers(0, P, 4), rng.integers(0, P, 4)
        assert list(so.emul(a, b)) == _emul_ref(a, b)
        assert list(so.emul(a, so.einv(a))) == [1, 0, 0, 0]
    x = np.array([0, 1, 0, 0])                                   # X^4 = 11
    x2 = so.emul(x, x)
    assert list(so.emul(x2, x2)) == [11, 0, 0, 0]


def test_ntt_vs_naive_dft_and_roundtrip():
    rng = np.random.default_rng(2)
    for lg in (1, 3, 6, 9):
        n = 1 << lg
        a = rng.integers(0, P, n).astype(np.uint32)
        w = so.lib().so_root_of_unity(lg)
        naive = [sum(int(a[k]) * pow(w, j * k, P) for k in range(n)) % P for j in range(n)]
        got = so.ntt(a)
        assert list(got) == naive
        assert np.array_equal(so.ntt(got, inverse=True), a)
    a = rng.integers(0, P, 1 << 12).astype(np.uint32)
    assert np.array_equal(so.ntt(so.ntt(a), inverse=True), a)


def test_lde_is_the_same_polynomial_on_the_coset():
    rng = np.random.default_rng(3)
    n, lb = 64, 1
    ev = rng.integers(0, P, n).astype(np.uint32)
    coeffs, out = so.lde(ev, lb)
    w2 = so.lib().so_root_of_unity(7)
    for j in (0, 1, 5, 64, 127):
        x = 31 * pow(w2, j, P) % P
        assert out[j] == sum(int(c) * pow(x, k, P) for k, c in enumerate(coeffs)) % P
    w = so.lib().so_root_of_unity(6)
    for i in (0, 3, 63):                                         # interpolant agrees with the trace on H
        assert ev[i] == sum(int(c) * pow(w, i * k, P) for k, c in enumerate(coeffs)) % P


def test_poseidon2_structure():
    ext, inn, diag = so.constants()
    assert (ext < P).all() and (inn < P).all() and len(np.unique(ext)) == 96
    assert list(diag) == [P - 2] + [1 << i for i in range(11)]
    z = so.permute(np.zeros(12))
    assert z.any() and (z < P).all()
    a = so.permute(np.arange(12))
    b = so.permute(np.arange(12) + np.eye(12, dtype=np.int64)[0])
    assert (a != b).sum() >= 10                                  # avalanche
    # sponge = overwrite-mode absorb of 8-element chunks
    x = np.arange(1, 20) % P
    s = np.zeros(12, np.uint32)
    for off in range(0, 19, 8):
        chunk = x[off:off + 8]
        s[:len(chunk)] = chunk
        s = so.permute(s)
    assert np.array_equal(so.hash_elems(x), s[:4])
    l, r = np.array([1, 2, 3, 4]), np.array([5, 6, 7, 8])
    assert np.array_equal(so.compress(l, r), so.permute(np.concatenate([l, r, np.zeros(4)]))[:4])
    assert not np.array_equal(so.compress(l, r), so.compress(r, l))


def test_poseidon2_matches_python_reference():
    """Independent python-int evaluation of the permutation from the published constants."""
    ext, inn, diag = so.constants()
    M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]

    def ext_lin(s):
        y = []
        for k in range(0, 12, 4):
            y += [sum(M4[i][j] * s[k + j] for j in range(4)) % P for i in range(4)]
        sums = [(y[j] + y[4 + j] + y[8 + j]) % P for j in range(4)]
        return [(y[k] + sums[k & 3]) % P for k in range(12)]

    def int_lin(s):
        t = sum(s) % P
        return [(t + s[i] * int(diag[i])) % P for i in range(12)]

    s = [int(v) for v in range(100, 112)]
    want = so.permute(np.array(s))
    s = ext_lin(s)
    for r in range(4):
        s = ext_lin([pow((s[i] + int(ext[r][i])) % P, 7, P) for i in range(12)])
    for r in range(22):
        s[0] = pow((s[0] + int(inn[r])) % P, 7, P)
        s = int_lin(s)
    for r in range(4, 8):
        s = ext_lin([pow((s[i] + int(ext[r][i])) % P, 7, P) for i in range(12)])
    assert s == [int(v) for v in want]


def test_merkle_layers_and_paths():
    rng = np.random.default_rng(4)
    w, n = 5, 16
    mat = rng.integers(0, P, (w, n)).astype(np.uint32)
    root, layers = so.merkle(mat, want_layers=True)
    leaf = layers[:4 * n].reshape(n, 4)
    for j in (0, 7, 15):
        assert np.array_equal(leaf[j], so.hash_elems(mat[:, j]))
    # walk a path
    j, off, m = 11, 0, n
    node = leaf[j]
    while m > 1:
        sib = layers[off + 4 * (j ^ 1): off + 4 * (j ^ 1) + 4]
        node = so.compress(node, sib) if j % 2 == 0 else so.compress(sib, node)
        off += 4 * m; m //= 2; j //= 2
    assert np.array_equal(node, root)


# column map of the main trace (AIR v3: oracle/stark_oracle.cpp)
C_PC, C_OP, C_FA, C_LIMB, C_STATE, C_WR, C_SELB, C_SELC, C_XB, C_XC, C_Y, C_K, C_OPC, C_RC, C_S, C_C0, C_D0, C_DL0, C_NE, C_IV, C_TK = \
    1, 4, 5, 9, 57, 73, 88, 103, 118, 121, 124, 127, 134, 135, 139, 141, 143, 146, 147, 148, 151
C_K2, C_NZ, C_IVZ, C_FLAG, C_FX, C_K3, C_B0, C_RC2, C_G, C_SB, C_K4, C_Q = 152, 156, 157, 158, 159, 160, 162, 163, 167, 168, 169, 171


def Z(m, l):
    """z_l, the row's first range-checked pair of limbs: since AIR v6 it has no columns of its own, it IS its chunks."""
    return m[C_RC + 2 * l].astype(np.int64) + 1024 * m[C_RC + 2 * l + 1].astype(np.int64)

K_ADD, K_ADDI, K_BRE, K_JAL, K_OTH, K_HALT, K_PAD, K_SUB, K_BRU, K_SE, K_SU, K_JALR, K_OJ, K_CMN, K_CMZ = range(15)
K_BNE = K_BRE                                                    # BEQ / BNE share a class: the family's comparison with either polarity
KCOL = [C_K + k for k in range(7)] + [C_K2 + k for k in range(4)] + [C_K3 + k for k in range(2)] + [C_K4 + k for k in range(2)]
W = so.W_MAIN
OPCLASS = {0x00: K_ADD, 0x08: K_ADDI, 0x40: K_BRE, 0x41: K_BRE, 0x48: K_JAL, 0x01: K_SUB, 0x44: K_BRU, 0x45: K_BRU, 0x24: K_SE, 0x25: K_SE, 0x20: K_SU, 0x21: K_SU, 0x22: K_SU, 0x23: K_SU, 0x49: K_JALR, 0x42: K_BRU, 0x43: K_BRU, 0x26: K_CMN, 0x28: K_CMN, 0x27: K_CMZ}


def test_main_trace_columns_and_commit():
    blob = spec.fib_endless_program().to_bytes()
    n = 50                                                       # not a power of two: 14 padding rows
    rows = oracle.run(blob, max_cycles=n, enable_execution_trace=True).rows
    pub = so.public_inputs(n, blob)
    m = so.main_trace(rows, pub)
    assert m.shape == (172, 64) and (m < P).all()
    assert list(m[0]) == list(range(64))                         # the cycle column keeps counting through the padding
    assert np.array_equal(m[1][:n], rows["pc"] & 0xFFFFF) and not m[3].any()
    assert np.array_equal(m[C_OP][:n], rows["instruction"] & 0x7F)
    assert np.array_equal(m[C_LIMB + 3 * 4][:n], rows["registers"][:, 4] & 0xFFFFF)
    cls = m[KCOL]
    assert (cls.sum(axis=0) == 1).all()
    assert cls[K_PAD][n:].all() and cls[K_HALT][n - 1] == 1 and cls[K_HALT].sum() == 1 and not cls[K_PAD][:n].any()
    ops = rows["instruction"][:n - 1] & 0x7F
    for k, code in ((K_ADD, 0x00), (K_ADDI, 0x08), (K_BNE, 0x41), (K_JAL, 0x48)):
        assert np.array_equal(cls[k][:n - 1].astype(bool), ops == code)
    assert not cls[K_OTH].any() and not cls[K_SUB:].any()       # the fib loop is made of four constrained opcodes only
    # wr = one-hot of rd on writing rows; y = the value the next row shows in rd
    for i in range(n - 1):
        w = rows["instruction"][i]
        rd = (int(w) >> 7) & 0xF
        wr = m[C_WR:C_WR + 15, i]
        if cls[K_BNE][i] or rd == 0:
            assert not wr.any()
        else:
            assert wr.sum() == 1 and wr[rd - 1] == 1
            v = int(rows["registers"][i + 1, rd])
            assert [int(m[C_Y + l, i]) for l in range(3)] == [v & 0xFFFFF, (v >> 20) & 0xFFFFF, v >> 40]
    # opclass = class of the instruction WORD on every row (halt and padding rows included); the range chunks split y's two low limbs
    opc = np.array([OPCLASS.get(int(o), K_OTH) for o in m[C_OP]])
    assert np.array_equal(m[C_OPC], opc)
    assert (m[C_RC:C_RC + 4] < 1024).all() and not m[C_RC2:C_RC2 + 4].any() and not m[C_G].any() and not m[C_SB].any()   # the second range-checked pair serves ordered comparisons (and the high bits of what "other" rows write)
    wrote = cls[K_ADD] | cls[K_ADDI] | cls[K_JAL]
    assert np.array_equal(Z(m, 0)[wrote == 1], m[C_Y][wrote == 1]) and np.array_equal(Z(m, 1)[wrote == 1], m[C_Y + 1][wrote == 1]) and not Z(m, 0)[wrote == 0].any()      # z = the written limbs; zero on BNE / halt / pad rows
    # padding rows repeat the state of the last executed row
    assert (m[C_LIMB:C_STATE + 16, n:] == m[C_LIMB:C_STATE + 16, n - 1:n]).all() and (m[C_PC:C_PC + 3, n:] == m[C_PC:C_PC + 3, n - 1:n]).all()
    root, L = so.commit_trace(rows, 1, want_lde=True, pub=pub)
    # what is committed: the logical matrix minus the columns that are identically zero (R0's limbs, the 16 storage states of the default
    # mode, since v6 the class column "other, jumps", which only the deferred mode uses), packed: 152 columns, whole blocks of 8, no padding
    kept = [c for c in range(172) if not (C_LIMB <= c < C_LIMB + 3 or C_STATE <= c < C_STATE + 16 or c == C_K3 + 1)]
    assert not m[C_LIMB:C_LIMB + 3].any() and not m[C_STATE:C_STATE + 16].any() and len(kept) == 152 and not m[C_K3 + 1].any()
    mc = so.to_committed(m)
    assert mc.shape == (152, 64) and np.array_equal(mc, m[kept])
    assert L.shape == (152, 128)
    assert np.array_equal(so.merkle(L), root)
    coeffs, col = so.lde(m[0], 1)
    assert np.array_equal(col, L[0])
    coeffs, col = so.lde(m[C_WR], 1)
    assert np.array_equal(col, L[C_WR - 19])
    # deferred mode keeps the storage states (only R0's limbs and state are left out): 168 columns
    rows_d = oracle.run(blob, max_cycles=n, enable_execution_trace=True, enable_deferred_model=True).rows
    pub_d = so.public_inputs(n, blob, deferred=True)
    m_d = so.main_trace(rows_d, pub_d)
    kept_d = [c for c in range(172) if not (C_LIMB <= c < C_LIMB + 3 or c == C_STATE)]
    mc_d = so.to_committed(m_d, deferred=True)
    assert mc_d.shape == (168, 64) and np.array_equal(mc_d, m_d[kept_d]) and m_d[C_STATE + 1:C_STATE + 16].any()
    assert so.commit_trace(rows_d, 1, want_lde=True, pub=pub_d)[1].shape == (168, 128)


def test_main_trace_of_the_opcode_families():
    """AIR v3: SUB, SLTU / SGEU, SEQ / SNE, BEQ / BNE, BLTU / BGEU rows of spec.compare_loop_program: class = the word's family, the
    difference limbs with their borrows, the comparison flag, its polarity, the value written and the branch decision — against the VM's
    own rows (next pc, next register value)."""
    blob = spec.compare_loop_program().to_bytes()
    n = 600
    rows = oracle.run(blob, max_cycles=n, enable_execution_trace=True).rows
    pub = so.public_inputs(n, blob)
    m = so.main_trace(rows, pub)
    cls = m[KCOL]
    assert (cls.sum(axis=0) == 1).all() and not cls[K_OTH].any()            # every opcode of this program is constrained
    M40 = (1 << 40) - 1
    seen = set()
    for i in range(n - 1):
        w = int(rows["instruction"][i]); op = w & 0x7F; k = OPCLASS[op]
        assert cls[k][i] == 1 and m[C_OPC, i] == k
        fa, fb, fc = (w >> 7) & 0xF, (w >> 11) & 0xF, (w >> 15) & 0xF
        reg = [int(v) for v in rows["registers"][i]]
        z = int(Z(m, 0)[i]) | (int(Z(m, 1)[i]) << 20)
        if k in (K_SUB, K_SU):
            assert z == (reg[fb] - reg[fc]) & M40 and m[C_C0 + 1, i] == int((reg[fb] & M40) < (reg[fc] & M40))
        if k == K_BRU:
            assert z == (reg[fa] - reg[fb]) & M40 and m[C_C0 + 1, i] == int((reg[fa] & M40) < (reg[fb] & M40))
        if k in (K_BRE, K_SE):
            a, b = (reg[fa], reg[fb]) if k == K_BRE else (reg[fb], reg[fc])
            assert m[C_FLAG, i] == int(a == b)
        if k in (K_BRU, K_SU):
            assert m[C_FLAG, i] == m[C_C0 + 1, i]
        if k in (K_BRE, K_BRU, K_SE, K_SU):
            assert m[C_FX, i] == int(m[C_FLAG, i]) ^ (op & 1)
        if k in (K_BRE, K_BRU):
            taken = int(rows["pc"][i + 1]) != int(rows["pc"][i]) + 4
            assert m[C_TK, i] == int(taken) == m[C_FX, i] and not m[C_WR:C_WR + 15, i].any()
            seen.add((op, taken))
        else:
            assert m[C_TK, i] == 0
        if k in (K_SE, K_SU):
            assert [int(m[C_Y + l, i]) for l in range(3)] == [int(m[C_FX, i]), 0, 0] and int(rows["registers"][i + 1, fa]) == m[C_FX, i]
            seen.add((op, int(m[C_FX, i])))
        if k == K_SUB:
            assert int(rows["registers"][i + 1, fa]) == z == int(m[C_Y, i]) | (int(m[C_Y + 1, i]) << 20) and m[C_Y + 2, i] == 0
            seen.add((op, int(m[C_C0 + 1, i])))
    # every comparison came out both ways, every branch was both taken and not taken, SUB with and without a borrow out of 40 bits
    assert seen == {(op, v) for op in (0x01, 0x20, 0x21, 0x24, 0x25, 0x40, 0x41, 0x44, 0x45) for v in (0, 1)}


def _s40(v):
    v &= (1 << 40) - 1
    return v - (1 << 40) if v >> 39 else v


def test_main_trace_of_the_signed_comparisons():
    """AIR v5: SLT / SGE / BLT / BGE rows of spec.signed_loop_program (and SLTU / SGEU / BLTU for contrast): class = the unsigned family's,
    the word's variant bit g (op = base + 2 g + pol), the sign bits, the biased high limbs as the second range-checked pair u, the borrow of
    the biased difference = Value40::signed_lt (value.rs:710-716), polarity, the value written / the branch decision — against the VM's rows."""
    blob = spec.signed_loop_program().to_bytes()
    n = 700
    rows = oracle.run(blob, max_cycles=n, enable_execution_trace=True).rows
    pub = so.public_inputs(n, blob)
    m = so.main_trace(rows, pub)
    cls = m[KCOL]
    assert (cls.sum(axis=0) == 1).all()
    M40 = (1 << 40) - 1
    seen = set()
    for i in range(n - 1):
        w = int(rows["instruction"][i]); op = w & 0x7F; k = OPCLASS.get(op, K_OTH)
        assert cls[k][i] == 1 and m[C_OPC, i] == k
        assert m[C_G, i] == int(op in (0x22, 0x23, 0x44, 0x45))
        if k not in (K_SU, K_BRU):
            assert not m[C_RC2:C_RC2 + 4, i].any() and m[C_SB, i] == 0
            continue
        fa, fb, fc = (w >> 7) & 0xF, (w >> 11) & 0xF, (w >> 15) & 0xF
        reg = [int(v) for v in rows["registers"][i]]
        a, b = (reg[fb] & M40, reg[fc] & M40) if k == K_SU else (reg[fa] & M40, reg[fb] & M40)
        signed = op in (0x22, 0x23, 0x42, 0x43)
        sa, sb = (a >> 39, b >> 39) if signed else (0, 0)
        assert (int(m[C_B0, i]), int(m[C_SB, i])) == (sa, sb)
        u = [int(m[C_RC2, i]) + 1024 * int(m[C_RC2 + 1, i]), int(m[C_RC2 + 2, i]) + 1024 * int(m[C_RC2 + 3, i])]
        bias = (1 << 19) if signed else 0
        assert u == [(a >> 20) + bias - (sa << 20), (b >> 20) + bias - (sb << 20)] and (m[C_RC2:C_RC2 + 4, i] < 1024).all()
        lt = int(_s40(a) < _s40(b)) if signed else int(a < b)
        assert m[C_C0 + 1, i] == lt == m[C_FLAG, i] and m[C_FX, i] == lt ^ (op & 1)
        z = int(Z(m, 0)[i]) | (int(Z(m, 1)[i]) << 20)
        assert z == ((a ^ (bias << 20)) - (b ^ (bias << 20))) & M40                # the difference of the biased values
        if k == K_BRU:
            taken = int(rows["pc"][i + 1]) != int(rows["pc"][i]) + 4
            assert m[C_TK, i] == int(taken) == m[C_FX, i] and not m[C_WR:C_WR + 15, i].any()
            seen.add((op, taken, sa, sb))
        else:
            assert int(rows["registers"][i + 1, fa]) == m[C_FX, i] and [int(m[C_Y + l, i]) for l in range(3)] == [int(m[C_FX, i]), 0, 0]
            seen.add((op, int(m[C_FX, i]), sa, sb))
    for op in (0x22, 0x23, 0x42, 0x43):                           # every signed comparison came out both ways, on every combination of operand signs it can
        assert {(o, v) for (o, v, _, _) in seen if o == op} == {(op, 0), (op, 1)}
    assert {(x, y) for (o, _, x, y) in seen if o in (0x22, 0x23)} == {(0, 0), (1, 1), (1, 0), (0, 1)}
    assert {(x, y) for (o, _, x, y) in seen if o in (0x42, 0x43)} == {(0, 0), (1, 1), (1, 0), (0, 1)}
    assert so.verify(so.prove(rows, pub)) == 0


def test_wrong_execution_of_the_signed_comparisons_is_rejected():
    """AIR v5 on spec.signed_loop_program: a signed comparison written the wrong way round, a signed branch going the other way (taken and
    not taken), forged sign bits (with the biased limb moved along: it leaves its range), a forged borrow, an unsigned comparison run as a
    signed one — rejected at the constraint check."""
    rows0, pub = _run(700, "sgn")
    ops = rows0["instruction"] & 0x7F
    rd_of = (rows0["instruction"] >> 7) & 0xF
    m0 = so.main_trace(rows0, pub)

    def rejected(rows):
        return so.verify(so.prove(rows, pub)) == 10
    for op in (0x22, 0x23):
        for want in (0, 1):
            ks = [int(k) for k in np.nonzero(ops[:-1] == op)[0] if int(rows0["registers"][k + 1, rd_of[k]]) == want]
            k = ks[len(ks) // 2]; rd = int(rd_of[k])
            later = np.nonzero((rd_of[k + 1:] == rd) & ~np.isin(ops[k + 1:], (0x42, 0x43, 0x44, 0x45)))[0]
            hi = k + 2 + int(later[0]) if len(later) else len(rows0)
            rows = rows0.copy(); rows["registers"][k + 1:hi, rd] ^= 1
            assert rejected(rows), (hex(op), want)
    for op in (0x42, 0x43):
        for want_taken in (False, True):
            ks = [int(k) for k in np.nonzero(ops[:-1] == op)[0] if (int(rows0["pc"][k + 1]) != int(rows0["pc"][k]) + 4) == want_taken]
            k = ks[len(ks) // 2]
            w = int(rows0["instruction"][k]); imm = (w >> 15) - (1 << 17 if w >> 31 else 0)
            rows = rows0.copy(); rows["pc"][k + 1] = int(rows0["pc"][k]) + (4 if want_taken else imm)
            assert rejected(rows), (hex(op), want_taken)

    def bad(edit):
        m = m0.copy(); edit(m)
        return so.verify(so.prove_matrix(m, pub)) == 10
    # rows whose operands have different signs: the sign bits decide
    ksl = next(int(k) for k in np.nonzero(ops == 0x22)[0] if m0[C_B0, k] != m0[C_SB, k])
    kbl = next(int(k) for k in np.nonzero(ops == 0x42)[0] if m0[C_B0, k] != m0[C_SB, k])

    def flip_sign(m, k, col, rc):                                # the sign bit flipped, the biased limb following it: u leaves [0, 2^20), a chunk leaves the table
        sgn = int(m[col, k]); d = (1 if sgn else -1) << 20
        m[col, k] = 1 - sgn
        m[rc + 1, k] = (int(m[rc + 1, k]) + d // 1024) % P
    assert bad(lambda m: flip_sign(m, ksl, C_B0, C_RC2)) and bad(lambda m: flip_sign(m, ksl, C_SB, C_RC2 + 2))
    assert bad(lambda m: flip_sign(m, kbl, C_B0, C_RC2)) and bad(lambda m: flip_sign(m, kbl, C_SB, C_RC2 + 2))

    def flip_sign_and_borrow(m, k):                              # ... and the borrow with it, so that the difference constraint holds again: the range lookup still refuses u
        flip_sign(m, k, C_B0, C_RC2)
        c1 = int(m[C_C0 + 1, k]); m[C_C0 + 1, k] = 1 - c1; m[C_FLAG, k] = 1 - c1; m[C_FX, k] = 1 - int(m[C_FX, k])
    assert bad(lambda m: flip_sign_and_borrow(m, ksl))
    # the sign bit flipped with u kept in range: the biased-limb constraint no longer holds
    assert bad(lambda m: m.__setitem__((C_B0, ksl), 1 - int(m[C_B0, ksl])))
    assert bad(lambda m: m.__setitem__((C_SB, kbl), 1 - int(m[C_SB, kbl])))
    # an unsigned comparison claiming sign bits (g says unsigned: the bias is absent, a sign bit of 1 pushes u below zero)
    ku = next(int(k) for k in np.nonzero(ops == 0x20)[0] if (int(m0[C_XB + 1, k]) >> 19))
    assert bad(lambda m: flip_sign(m, ku, C_B0, C_RC2))
    # the variant bit forged (SLT run as SLTU): the ROM tuple is not the program's
    assert bad(lambda m: m.__setitem__((C_G, ksl), 0))
    # a forged flag / polarity / decision on a signed row
    assert bad(lambda m: m.__setitem__((C_FLAG, ksl), 1 - int(m[C_FLAG, ksl])))
    assert bad(lambda m: m.__setitem__((C_FX, kbl), 1 - int(m[C_FX, kbl])))
    assert bad(lambda m: m.__setitem__((C_TK, kbl), 1 - int(m[C_TK, kbl])))


def test_conditional_moves():
    """AIR v6: CMOV / CMOVNZ / CMOVZ rows of spec.cmov_loop_program — class of the word, nz = [rs2 != 0] over the raw 64 bits (stated on every
    row), q = the condition holds, the write selector (rd iff q, nothing for rd = r0), y = rs1's raw limbs — against the VM's rows; the range
    check of the bits above 40 that "other" rows write (the sign-extended byte load); and every way of forging a conditional move rejected."""
    rows0, pub = _run(400, "cmov")
    m = so.main_trace(rows0, pub)
    n = len(rows0)
    cls = m[KCOL]
    ops = rows0["instruction"] & 0x7F
    assert (cls.sum(axis=0) == 1).all()
    seen = set()
    for i in range(n - 1):
        w = int(rows0["instruction"][i]); op = w & 0x7F; k = OPCLASS.get(op, K_OTH)
        assert cls[k][i] == 1
        fa, fb, fc = (w >> 7) & 0xF, (w >> 11) & 0xF, (w >> 15) & 0xF
        reg = [int(v) for v in rows0["registers"][i]]
        if op not in (0x40, 0x41, 0x42, 0x43, 0x44, 0x45, 0x38, 0x39, 0x3A, 0x3B):                   # R- / I-type words: xc = reg[field c] (an immediate's bits for I-type: any register)
            assert m[C_NZ, i] == int(reg[fc] != 0)
        if k in (K_CMN, K_CMZ):
            cond = (reg[fc] != 0) == (k == K_CMN)
            assert m[C_Q, i] == int(cond)
            assert [int(m[C_Y + l, i]) for l in range(3)] == [reg[fb] & 0xFFFFF, (reg[fb] >> 20) & 0xFFFFF, reg[fb] >> 40]
            wr = m[C_WR:C_WR + 15, i]
            if cond and fa: assert wr.sum() == 1 and wr[fa - 1] == 1 and int(rows0["registers"][i + 1, fa]) == reg[fb]
            else: assert not wr.any() and (rows0["registers"][i + 1] == rows0["registers"][i]).all()
            seen.add((op, cond, fa == 0, reg[fc] >> 40 != 0, reg[fb] >> 40 != 0))
        else:
            assert m[C_Q, i] == 0
        if k == K_OTH:                                              # the bits above 40 of the written value, as three chunks (the third below 16)
            y2 = int(m[C_Y + 2, i])
            assert [int(v) for v in m[C_RC2:C_RC2 + 4, i]] == [y2 & 1023, (y2 >> 10) & 1023, y2 >> 20, 64 * (y2 >> 20)]
            if op == 0x30: assert y2 == 0xFFFFFF
    assert {(o, c) for (o, c, _, _, _) in seen} == {(o, c) for o in (0x26, 0x27, 0x28) for c in (False, True)}
    assert any(z for (_, _, z, _, _) in seen) and any(h for (_, _, _, h, _) in seen) and any(h for (_, _, _, _, h) in seen)
    assert so.verify(so.prove(rows0, pub)) == 0

    def rejected(rows):
        return so.verify(so.prove(rows, pub)) == 10
    rd_of = (rows0["instruction"] >> 7) & 0xF
    for op in (0x26, 0x27, 0x28):
        for want in (False, True):                                # a move that happened is undone; one that did not happen is made
            ks = [int(k) for k in np.nonzero((ops[:-1] == op) & (rd_of[:-1] != 0))[0] if bool(m[C_Q, k]) == want]
            k = ks[len(ks) // 2]; rd = int(rd_of[k]); src = (int(rows0["instruction"][k]) >> 11) & 0xF
            later = np.nonzero(rd_of[k + 1:] == rd)[0]
            hi = k + 2 + int(later[0]) if len(later) else n
            rows = rows0.copy()
            rows["registers"][k + 1:hi, rd] = rows0["registers"][k, rd] if want else rows0["registers"][k, src]
            assert rejected(rows), (hex(op), want)
    # a move of another value than rs1's (one limb off, the bits above 40 dropped)
    k = next(int(k) for k in np.nonzero(ops == 0x28)[0] if m[C_Q, k] and m[C_Y + 2, k])
    rd = int(rd_of[k]); later = np.nonzero(rd_of[k + 1:] == rd)[0]; hi = k + 2 + int(later[0])
    for edit in (lambda v: v + 1, lambda v: v & ((1 << 40) - 1)):
        rows = rows0.copy(); rows["registers"][k + 1:hi, rd] = edit(int(rows0["registers"][k + 1, rd]))
        assert rejected(rows)

    def bad(edit):
        mm = m.copy(); edit(mm)
        return so.verify(so.prove_matrix(mm, pub)) == 10
    kq = next(int(k) for k in np.nonzero((ops == 0x26) & (rd_of == 11))[0] if m[C_Q, k]); kn = next(int(k) for k in np.nonzero((ops == 0x26) & (rd_of == 11))[0] if not m[C_Q, k])
    assert bad(lambda mm: mm.__setitem__((C_Q, kq), 0)) and bad(lambda mm: mm.__setitem__((C_Q, kn), 1))          # q is not the condition
    assert bad(lambda mm: mm.__setitem__((C_NZ, kq), 0)) and bad(lambda mm: mm.__setitem__((C_NZ, kn), 1))        # nz is not [rs2 != 0]
    assert bad(lambda mm: (mm.__setitem__((C_NZ, kn), 1), mm.__setitem__((C_IVZ, kn), 1)))
    assert bad(lambda mm: (mm.__setitem__((C_WR + 10, kq), 0), mm.__setitem__((C_WR + 11, kq), 1)))               # another register than rd is flagged written
    assert bad(lambda mm: mm.__setitem__((C_WR + 10, kn), 1))                                                     # a write although the condition fails
    assert bad(lambda mm: (mm.__setitem__((C_K4, kq), 0), mm.__setitem__((C_K4 + 1, kq), 1)))                     # CMOV run as CMOVZ
    assert bad(lambda mm: (mm.__setitem__((C_K4, kq), 0), mm.__setitem__((C_K + K_OTH, kq), 1)))                  # ... or hiding as "other"
    # an "other" row whose written bits above 40 are out of range: the third chunk must be below 16 (64 x it is looked up too)
    kl = int(np.nonzero(ops == 0x30)[0][0])

    def y2_out_of_range(mm):
        mm[C_Y + 2, kl] = (int(mm[C_Y + 2, kl]) + (1 << 24)) % P; mm[C_RC2 + 2, kl] = int(mm[C_RC2 + 2, kl]) + 16; mm[C_RC2 + 3, kl] = 64 * int(mm[C_RC2 + 2, kl])
        mm[C_LIMB + 3 * 7 + 2, kl + 1:] = mm[C_Y + 2, kl]
    assert bad(y2_out_of_range)


def test_cpu_commit_port_matches_the_oracle():
    """bench_cpu/cpu_commit_port.cpp (the multi-threaded Montgomery port bench.py times as the CPU figure of the commit stage) computes
    the same root as the naive oracle."""
    for prog, n in ((spec.fib_endless_program(), 300), (spec.sha256_chain_program(), 1024)):
        blob = prog.to_bytes()
        rows = oracle.run(blob, max_cycles=n, enable_execution_trace=True).rows
        pub = so.public_inputs(len(rows), blob)
        root, t_lde, t_merkle = cpu_port.commit_port(so.to_committed(so.main_trace(rows, pub)), 3)
        assert np.array_equal(root, so.commit_trace(rows, 1, pub=pub)) and t_lde > 0 and t_merkle > 0


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_column_blocked_commit_is_the_commit(mode):
    """so_commit_trace_blocked (round 5: the 2^24-row golden root of tests/golden/config_roots.json — eight LDE columns at a time, one 12-word sponge state per leaf, std::threads
    over columns / leaves) gives the root of so_commit_trace (every LDE column in memory, hash_elems per leaf) in every mode (152 / 168 / 160 / 264 committed columns), ragged row
    counts, one thread and several."""
    for prog, n in ((spec.fib_endless_program(), 300), (spec.memory_loop_program(40), 1000), (spec.sha256_chain_program(), 513)):
        blob = prog.to_bytes()
        res = oracle.run(blob, max_cycles=n, enable_execution_trace=True, enable_deferred_model=mode == 1)
        pub = so.public_inputs(len(res.rows), blob, [], list(res.outputs), (res.halt_kind, res.halt_code), deferred=mode == 1, io_mode=mode == 2, mem_mode=mode == 3, wide_mode=mode == 4)
        want = so.commit_trace(res.rows, 1, pub=pub)
        for threads in (1, 3):
            assert np.array_equal(so.commit_trace_blocked(res.rows, 1, pub=pub, threads=threads), want), (mode, n, threads)


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_lean_prover_equals_the_plain_one(mode):
    """so::prove_lean (round 6: the whole-proof golden at BASELINE configs[2]'s own size, 2^24 rows — no committed copy of the matrix, openings from coefficients re-derived
    out of the LDE's even positions, std::threads over columns / leaves / coset points) writes the proof of so::prove word for word: every mode (152 / 168 / 160 / 264 committed
    columns; the I/O and memory sections, the touched cells), ragged row counts, a self-halting run, one thread and several, the second parameter set."""
    cases = [(spec.fib_endless_program(), 300, {}), (spec.fib_program(12), None, {}), (spec.fib_endless_program(), 2048, {"num_queries": 84, "pow_bits": 16})]
    if mode != 1:
        cases.append((spec.memory_loop_program(40), 700, {}))
    if mode == 4:
        cases.append((spec.wide_loop_program(), 300, {}))
    for prog, n, params in cases:
        blob = prog.to_bytes()
        res = oracle.run(blob, max_cycles=n or 1_000_000, enable_execution_trace=True, enable_deferred_model=mode == 1)
        pub = so.public_inputs(len(res.rows), blob, [], list(res.outputs), (res.halt_kind, res.halt_code), deferred=mode == 1, io_mode=mode == 2, mem_mode=mode == 3, wide_mode=mode == 4, **params)
        want = so.prove(res.rows, pub)
        assert so.verify(want, pub) == 0
        for threads in (1, 3):
            assert np.array_equal(so.prove_lean(res.rows, pub, threads=threads), want), (mode, n, threads)


def test_air_holds_row_by_row_on_honest_traces():
    """Every constraint vanishes on every (row, next row) pair of an honest main trace: evaluated here with the row selectors a
    verifier would use ON the trace domain (is_first = [i == 0], is_last = [i == n_real - 1], is_trans = [i != N - 1])."""
    for prog, n, cfg in ((spec.fib_endless_program(), 50, {}), (spec.sha256_chain_program(), 200, {}), (spec.fib_program(12), None, {}),
                         (spec.fib_endless_program(), 40, {"enable_deferred_model": True}), (spec.compare_loop_program(), 600, {}),
                         (spec.compare_loop_program(), 100, {"enable_deferred_model": True}), (spec.call_loop_program(), 500, {}),
                         (spec.call_loop_program(), 120, {"enable_deferred_model": True}),
                         (spec.signed_loop_program(), 300, {}), (spec.signed_loop_program(), 100, {"enable_deferred_model": True}), (spec.cmov_loop_program(), 300, {}),
                         (spec.cmov_loop_program(), 100, {"enable_deferred_model": True})):
        blob = prog.to_bytes()
        res = oracle.run(blob, max_cycles=n or 1_000_000, enable_execution_trace=True, **cfg)
        rows = res.rows
        pub = so.public_inputs(len(rows), blob, deferred=bool(cfg))
        m = so.main_trace(rows, pub)
        N = m.shape[1]
        alpha = np.array([7, 11, 13, 17], np.uint32)
        # the lookup side for some challenges: aux trace (helper columns, running sum), alpha / lambda powers / T / N; the running sum closes
        # over the cycle (row N - 1 -> row 0) because the row side of the LogUp identity equals the table side T
        aux, lk, rom_mult, rc_mult = so.lookup_setup(m, pub, [5, 6, 7, 8], [9, 10, 11, 12])
        assert int(rom_mult.sum()) == N and int(rc_mult.sum()) == 8 * N       # every row looks its instruction up once, its eight chunks once each
        for i in range(N):
            out = so.constraints_eval(m[:, i], m[:, (i + 1) % N], aux[:, i], aux[:, (i + 1) % N], lk, int(i == 0), int(i == len(rows) - 1), int(i != N - 1), pub, alpha)
            assert not out.any(), (i, len(rows))
        if not cfg:
            # a chunk outside the table, or a row whose tuple is not a ROM row, breaks the identity: the sum no longer closes
            bad = m.copy(); bad[C_RC + 1, 3] = 1024
            aux2, lk2, _, rc2 = so.lookup_setup(bad, pub, [5, 6, 7, 8], [9, 10, 11, 12])
            assert int(rc2.sum()) == 8 * N - 1
            assert any(so.constraints_eval(bad[:, i], bad[:, (i + 1) % N], aux2[:, i], aux2[:, (i + 1) % N], lk2, int(i == 0), int(i == len(rows) - 1), int(i != N - 1), pub, alpha).any()
                       for i in range(N))


# ---- stage B: prover + verifier ------------------------------------------------------------------------------------
def _prog(prog):
    return {"fib": spec.fib_endless_program, "sha": spec.sha256_chain_program, "cmp": spec.compare_loop_program, "call": spec.call_loop_program, "sgn": spec.signed_loop_program, "cmov": spec.cmov_loop_program}[prog]()


def _run(n, prog="fib", **cfg):
    blob = _prog(prog).to_bytes()
    res = oracle.run(blob, max_cycles=n, enable_execution_trace=True, **cfg)
    return res.rows, so.public_inputs(len(res.rows), blob, deferred=bool(cfg.get("enable_deferred_model")))


def _rows(n, prog="fib", **cfg):
    return _run(n, prog, **cfg)[0]


def _fri_schedule(log_n, log_final=3, log_arity=3):
    """Folds per committed FRI layer (oracle/stark_oracle.cpp: fri_schedule): layer 0 once, then 8-to-1, shorter at the end."""
    ks, log_m = [], log_n + 1
    while log_m > log_final:
        k = 1 if not ks else min(log_arity, log_m - log_final)
        ks.append(k)
        log_m -= k
    return ks


HDR = 21 + 2 * 68   # header words: parameters, public inputs, the two boundary states; then (format v5) the program and the lookup multiplicities
NQ = 50
WA, WC = so.W_AUX, so.W_COMMITTED                                # aux columns; committed main columns of a default-mode proof
WT = WC + WA


@pytest.mark.parametrize("n,prog", [(8, "fib"), (16, "fib"), (5, "fib"), (100, "fib"), (256, "fib"), (300, "sha"), (1024, "fib"), (600, "cmp"), (500, "call")])
def test_prove_verify_roundtrip(n, prog):
    rows, pub = _run(n, prog)
    pr = so.prove(rows, pub)
    log_n = so.padded_log_n(n)
    ks = _fri_schedule(log_n)                                                              # [1], [1,1], [1,2], [1,3,1], [1,3,2], [1,3,3], [1,3,3,1]
    lay = so.proof_layout(pr)
    blob = _prog(prog).to_bytes()
    assert lay["blob"] == blob and lay["n_rom"] == int.from_bytes(blob[16:20], "little") // 4 and lay["trace_root"] == HDR + 1 + (len(blob) + 1) // 2 + lay["n_rom"] + 1024
    fixed = lay["trace_root"] + 12 + (2 * WT + 4) * 4                                      # ... roots (trace, aux, quotient), openings of main + aux columns and the quotient
    assert pr[1] == 10 and pr[fixed] == len(ks)                                             # proof version, number of committed FRI layers
    assert int(pr[lay["rom_mult"]:lay["rc_mult"]].sum()) == 1 << log_n and int(pr[lay["rc_mult"]:lay["trace_root"]].sum()) == 8 << log_n   # multiplicities count every row
    depth = [log_n + 1 - sum(ks[:j + 1]) for j in range(len(ks))]                          # Merkle depth of each FRI tree
    per_query = 1 + 2 * (WC + 4 * (log_n + 1)) + 2 * (WA + 4 * (log_n + 1)) + 2 * (4 + 4 * (log_n + 1)) + sum(4 * (1 << k) + 4 * d for k, d in zip(ks, depth))
    assert len(pr) == fixed + 1 + 4 * len(ks) + 4 * 8 + 1 + NQ * per_query
    assert pr[0] == 0x46504B5A and pr[2] == log_n and pr[3] == WC and pr[4] == NQ and pr[6] == 12 and pr[7] == n
    assert (pr[2:] < P).all()
    assert so.verify(pr) == 0 and so.verify(pr, pub) == 0
    assert so.verify(pr[:-1]) != 0 and so.verify(np.concatenate([pr, [0]])) != 0           # length is checked
    rng = np.random.default_rng(n)
    for pos in list(range(7, HDR)) + [int(x) for x in rng.integers(HDR, len(pr), 40)]:     # any single-word change is rejected, public inputs included
        t = pr.copy()
        t[pos] = (int(t[pos]) + 1 + int(rng.integers(0, 1000))) % P
        if t[pos] != pr[pos]:
            assert so.verify(t) != 0, pos


def test_public_inputs_are_bound():
    """The header carries (rows, mode, entry pc, program digest, io digest); the transcript absorbs it before anything else, and a
    verifier that expects other public inputs rejects."""
    blob = spec.fib_program(12).to_bytes()
    res = oracle.run(blob, enable_execution_trace=True)
    pub = so.public_inputs(len(res.rows), blob, [], list(res.outputs), (res.halt_kind, res.halt_code))
    pr = so.prove(res.rows, pub)
    assert so.verify(pr, pub) == 0
    other_prog = so.public_inputs(len(res.rows), spec.fib_program(13).to_bytes(), [], list(res.outputs), (res.halt_kind, res.halt_code))
    other_out = so.public_inputs(len(res.rows), blob, [], [999], (res.halt_kind, res.halt_code))
    other_halt = so.public_inputs(len(res.rows), blob, [], list(res.outputs), (2, 0))
    for e in (other_prog, other_out, other_halt):
        assert so.verify(pr, e) == 6
    a, b = so.prove(res.rows, pub), so.prove(res.rows, other_out)                          # same trace, other claimed outputs: different challenges throughout
    t0 = so.proof_layout(a)["trace_root"]
    assert np.array_equal(a[t0:t0 + 4], b[t0:t0 + 4]) and not np.array_equal(a[t0 + 4:t0 + 8], b[t0 + 4:t0 + 8]) and not np.array_equal(a[t0 + 8:t0 + 12], b[t0 + 8:t0 + 12])


def test_proof_of_work_nonce():
    rows, pub = _run(32)
    pr = so.prove(rows, pub)
    log_n, ks = 5, _fri_schedule(5)
    at = so.proof_layout(pr)["trace_root"] + 12 + (2 * WT + 4) * 4 + 1 + 4 * len(ks) + 4 * 8
    t = pr.copy(); t[at] = (int(t[at]) + 1) % P
    assert so.verify(t) in (12, 20)                              # another nonce: the grinding check (or, if it happens to pass, the query indices) fails


def test_quotient_is_low_degree_and_fri_layers_fold():
    n = 64
    so.prove(*_run(n))
    Q = so.last_quotient(n)
    for i in range(4):                                                                     # degree < N: upper half of the coset-coefficients vanish
        c = so.ntt(Q[i], inverse=True)
        assert not c[n:].any()
    l0, l1 = so.last_fri_layer(0), so.last_fri_layer(1)
    assert l0.shape == (2 * n, 4) and l1.shape == (n, 4)
    # DEEP codeword has degree < N as well (each quotient (p(x)-p(z))/(x-z) drops one degree)
    for t in range(4):
        c = so.ntt(l0[:, t], inverse=True)
        assert not c[n:].any()


def test_invalid_traces_are_rejected():
    def code(mutate, n=64):
        rows, pub = _run(n)
        rows = rows.copy()
        mutate(rows)
        return so.verify(so.prove(rows, pub))
    assert code(lambda r: r["cycle"].__setitem__(30, 77)) == 10                    # cycle counter must increase by one
    # R0 is hard-wired zero: its limbs are not even committed (format v7) — a trace whose R0 holds 1 has operands the selectors cannot produce
    assert code(lambda r: r["registers"].__setitem__((slice(1, None), 0), 1)) == 10
    assert code(lambda r: r["registers"].__setitem__((slice(None), 0), 1)) == 10
    # default mode: every register is Normalized, the storage states are not committed either — a row that claims another state splits
    # its limbs differently and breaks the register's continuity
    assert code(lambda r: r["reg_state"].__setitem__((10, 3), 2)) == 10
    assert code(lambda r: r["reg_state"].__setitem__((10, 3), 1)) == 10
    # a run that does not START in the VM's initial state: the first state in the header (pinned to row 0 by the AIR) gives it away
    assert code(lambda r: r["cycle"].__iadd__(5)) == 7                             # first row must be cycle 0
    assert code(lambda r: r["registers"].__setitem__((0, 5), 3)) == 7              # registers start at zero (state.rs:55-71)
    assert code(lambda r: r["pc"].__iadd__(8)) == 7                                # row 0 is at the entry point


def test_boundary_states_are_pinned_to_the_trace():
    """Format v4: the header's first / last state are rows 0 and n_real - 1 of the committed trace — a header that claims other
    states fails the constraint check (the AIR's is_first / is_last constraints), on every one of the 136 words."""
    rows, pub = _run(100)
    pr = so.prove(rows, pub)
    m = so.main_trace(rows, pub)
    rc, first, last = so.verify_segment(pr, pub)
    assert rc == 0 and np.array_equal(first, m[so.STATE_COLS, 0]) and np.array_equal(last, m[so.STATE_COLS, 99])
    assert first[0] == 0 and last[0] == 99 and not first[4:].any()                 # cycle 0 .. 99; zero registers at the start
    for pos in range(21, 21 + 136):
        t = pr.copy(); t[pos] = (int(t[pos]) + 1) % P
        assert so.verify_segment(t, pub)[0] != 0, pos                              # (also changes the transcript: rejected either way)
    # and as constraints: a segment's last-state constraint fails for a wrong claimed last state
    alpha = np.array([3, 1, 4, 1], np.uint32)
    aux, lk, _, _ = so.lookup_setup(m, pub, [2, 7, 1, 8], [2, 8, 1, 8])
    ok = so.constraints_eval_states(m[:, 99], m[:, 100], aux[:, 99], aux[:, 100], lk, 0, 1, 1, pub, first, last, alpha)
    assert not ok.any()
    bad = last.copy(); bad[7] = (int(bad[7]) + 1) % P
    assert so.constraints_eval_states(m[:, 99], m[:, 100], aux[:, 99], aux[:, 100], lk, 0, 1, 1, pub, first, bad, alpha).any()
    badf = first.copy(); badf[2] ^= 1
    assert so.constraints_eval_states(m[:, 0], m[:, 1], aux[:, 0], aux[:, 1], lk, 1, 0, 1, pub, badf, last, alpha).any()


def _segments(n_total, seg_rows, prog="fib", deferred=False):
    """The run's rows cut into segments of `seg_rows` rows that overlap by one row; the last one takes what is left."""
    from zkir_amd import spec
    blob = (spec.fib_endless_program() if prog == "fib" else spec.sha256_chain_program()).to_bytes()
    res = oracle.run(blob, max_cycles=n_total, enable_execution_trace=True, enable_deferred_model=deferred)
    rows = res.rows
    cuts, a = [], 0
    while a < len(rows) - 1:
        b = min(a + seg_rows, len(rows))
        cuts.append((a, b)); a = b - 1
    full = so.public_inputs(len(rows), blob, [], list(res.outputs), (res.halt_kind, res.halt_code), deferred=deferred)
    proofs = []
    for a, b in cuts:
        pub = so.public_inputs(b - a, blob, [], list(res.outputs), (res.halt_kind, res.halt_code), deferred=deferred)
        pub.io[:] = full.io[:]                                                     # every segment carries the RUN's io digest
        proofs.append(so.prove(rows[a:b], pub))
    return rows, full, cuts, proofs


@pytest.mark.parametrize("n_total,seg", [(100, 40), (256, 64), (300, 128), (65, 33)])
def test_a_run_proven_in_segments(n_total, seg):
    """Multi-GPU proving: each row shard (plus the first row of the next) is proven on its own; the chain verifier
    accepts the segments as ONE run — first state initial, each segment starting where the previous ended, same program / io."""
    rows, full, cuts, proofs = _segments(n_total, seg)
    assert len(proofs) >= 2 and sum(b - a - 1 for a, b in cuts) + 1 == n_total
    assert so.verify_chain(proofs, full) == 0 and so.verify_chain(proofs) == 0
    assert so.verify(proofs[0]) == 0                                               # the first segment is also a valid whole-run proof of its rows
    assert so.verify(proofs[1]) == 7                                               # a later segment is not: it does not start in the initial state
    rc, f1, l1 = so.verify_segment(proofs[1])
    rc0, f0, l0 = so.verify_segment(proofs[0])
    assert rc == rc0 == 0 and np.array_equal(f1, l0) and f1[0] == cuts[1][0]       # linked states; cycle counter = first row of the segment
    # chain checks
    assert so.verify_chain(proofs[1:]) == 41                                       # does not start at the beginning
    assert so.verify_chain([proofs[0]] + proofs[2:] if len(proofs) > 2 else [proofs[0], proofs[0]]) == 42      # a gap / a repeat: states do not link
    assert so.verify_chain(proofs[::-1]) in (41, 42)
    wrong = so.PublicC.from_buffer_copy(full); wrong.n_real = full.n_real + 1
    assert so.verify_chain(proofs, wrong) == 44
    other = so.PublicC.from_buffer_copy(full); other.prog[0] ^= 1
    assert so.verify_chain(proofs, other) == 43
    t = proofs[1].copy(); t[-1] = (int(t[-1]) + 1) % P
    assert so.verify_chain([proofs[0], t] + proofs[2:]) // 1000 == 2               # segment 1 (index + 1 = 2) fails its own checks


def test_segments_of_a_different_run_do_not_splice():
    """A segment of ANOTHER program (or of the same program with other data) cannot be spliced into a chain: its first state is not
    the predecessor's last state, or its program digest differs."""
    rows, full, cuts, proofs = _segments(128, 64, "fib")
    rows2, full2, cuts2, proofs2 = _segments(128, 64, "sha")
    assert so.verify_chain([proofs[0], proofs2[1]]) == 43
    _, _, _, shifted = _segments(128, 48, "fib")                                   # same run, cut elsewhere: states do not line up
    assert so.verify_chain([proofs[0], shifted[1]]) == 42


def test_wrong_execution_is_rejected():
    """What the v0 AIR could not see (VERDICT r1 'missing' 1): a trace whose register VALUES or control flow do not follow the
    program.  Every mutation keeps the trace internally 'continuous' (the wrong value persists until the next write)."""
    rows0, pub = _run(64)
    ops = rows0["instruction"] & 0x7F

    def proof_of(rows):
        return so.verify(so.prove(rows, pub))
    # ADD r4, r1, r2 at some row k: its result, visible from row k+1 until r4 is written again, is off by one
    k = int(np.nonzero(ops == 0x00)[0][2])
    nxt = k + 1 + int(np.nonzero(ops[k + 1:] == 0x00)[0][0])
    rows = rows0.copy(); rows["registers"][k + 1:nxt + 1, 4] += 1
    assert proof_of(rows) == 10
    # ADDI r3, r3, -1: the counter does not decrement
    k = int(np.nonzero((ops == 0x08) & (((rows0["instruction"] >> 7) & 0xF) == 3))[0][3])
    nxt = k + 1 + int(np.nonzero((ops[k + 1:] == 0x08) & (((rows0["instruction"][k + 1:] >> 7) & 0xF) == 3))[0][0])
    rows = rows0.copy(); rows["registers"][k + 1:nxt + 1, 3] = rows["registers"][k, 3]
    assert proof_of(rows) == 10
    # a write lands in the wrong register (r5 instead of rd)
    k = int(np.nonzero(ops == 0x00)[0][1])
    rows = rows0.copy(); rows["registers"][k + 1:, 5] = 123
    assert proof_of(rows) == 10
    # BNE taken, but the next row continues at pc + 4 (and stays consistent afterwards: every later pc shifted the same way)
    k = int(np.nonzero(ops == 0x41)[0][1])
    rows = rows0.copy(); rows["pc"][k + 1] = rows["pc"][k] + 4
    assert proof_of(rows) == 10
    # JAL r0, -20 lands somewhere else
    rows1, pub1 = _run(64, "sha")
    k = int(np.nonzero((rows1["instruction"] & 0x7F) == 0x48)[0][0])
    rows = rows1.copy(); rows["pc"][k + 1] += 4
    assert so.verify(so.prove(rows, pub1)) == 10
    # one instruction word replaced by another opcode's while the registers still follow the original program: ADD -> SUB, a class
    # "other" row whose value the AIR does not constrain.  AIR v1 ACCEPTED this (VERDICT r2 missing #1: nothing tied the word at pc to the
    # program); with the instruction-ROM lookup the tuple (pc, word fields) of that row is not a row of the program's code table
    k = int(np.nonzero(ops == 0x00)[0][2])
    rows = rows0.copy(); rows["instruction"][k] = (int(rows["instruction"][k]) & ~0x7F) | 0x01
    assert proof_of(rows) == 10
    # ... the same with any other field of the word (rd, rs1, rs2 / the immediate), or a whole foreign instruction stream
    for shift in (7, 11, 15, 19, 31):
        rows = rows0.copy(); rows["instruction"][k] ^= np.uint32(1 << shift)
        assert proof_of(rows) != 0, shift
    blob2 = spec.fib_program(9).to_bytes()                       # another program's rows under this program's public inputs
    r2 = oracle.run(blob2, enable_execution_trace=True).rows
    assert so.verify(so.prove(r2, so.public_inputs(len(r2), spec.fib_endless_program().to_bytes()))) != 0


def test_wrong_execution_of_the_opcode_families_is_rejected():
    """AIR v3: the same for SUB, the unsigned and the equality comparisons and the four branches they drive: a wrong difference, a
    comparison written the wrong way round, a branch that goes the other way — each kept consistent afterwards, each rejected."""
    rows0, pub = _run(600, "cmp")
    ops = rows0["instruction"] & 0x7F
    rd_of = (rows0["instruction"] >> 7) & 0xF
    assert so.verify(so.prove(rows0, pub)) == 0

    def rejected(rows):
        return so.verify(so.prove(rows, pub)) == 10

    def until_rewritten(k, rd):                                  # rows k+1 .. (the next write of rd): where a forged value of rd shows
        later = np.nonzero((rd_of[k + 1:] == rd) & ~np.isin(ops[k + 1:], (0x40, 0x41, 0x44, 0x45)))[0]
        return k + 1, k + 2 + int(later[0]) if len(later) else len(rows0)
    for op, delta in ((0x01, 1), (0x01, 1 << 20), (0x20, None), (0x21, None), (0x24, None), (0x25, None)):
        k = int(np.nonzero(ops == op)[0][3]); rd = int(rd_of[k])
        lo, hi = until_rewritten(k, rd)
        rows = rows0.copy()
        if delta is None: rows["registers"][lo:hi, rd] ^= 1      # the comparison written the wrong way round
        else: rows["registers"][lo:hi, rd] = (rows["registers"][lo:hi, rd] + delta) & ((1 << 40) - 1)
        assert rejected(rows), hex(op)
    # SUB that forgets the wrap: a - b below zero written as a 64-bit two's complement (bits above 40 set)
    k = int(np.nonzero((ops == 0x01) & (rd_of == 5))[0][2]); lo, hi = until_rewritten(k, 5)
    rows = rows0.copy(); rows["registers"][lo:hi, 5] |= np.uint64(0xFFFFFF << 40)
    assert rejected(rows)
    # branches going the other way: the next pc is the other candidate (everything after it shifted along is then a different run: only
    # the one row is forged here, the ROM lookup and the pc constraint both see it)
    for op in (0x40, 0x41, 0x44, 0x45):
        for want_taken in (False, True):
            ks = [int(k) for k in np.nonzero(ops[:-1] == op)[0] if (int(rows0["pc"][k + 1]) != int(rows0["pc"][k]) + 4) == want_taken]
            k = ks[len(ks) // 2]
            w = int(rows0["instruction"][k]); imm = (w >> 15) - (1 << 17 if w >> 31 else 0)
            rows = rows0.copy(); rows["pc"][k + 1] = int(rows0["pc"][k]) + (4 if want_taken else imm)
            assert rejected(rows), (hex(op), want_taken)


def test_control_flow_of_every_opcode():
    """AIR v4 / v5 on spec.call_loop_program: JALR rows (link = pc + 4; next pc + the cleared bit = rs1 + sext(imm), even and odd sums, a
    negative immediate), class "other" rows (MUL, SLLI: pc + 4 enforced), signed branches (BLT / BGE: since v5 rows of the ordered-branch
    class, decided by the signed comparison, nothing written) — the columns against the VM's rows, and every way of bending the control
    flow rejected."""
    rows0, pub = _run(500, "call")
    m = so.main_trace(rows0, pub)
    ops = rows0["instruction"] & 0x7F
    cls = m[KCOL]
    assert (cls.sum(axis=0) == 1).all()
    n = len(rows0)
    seen = set()
    for i in range(n - 1):
        op = int(ops[i]); k = OPCLASS.get(op, K_OTH)
        assert cls[k][i] == 1
        w = int(rows0["instruction"][i]); fa, fb = (w >> 7) & 0xF, (w >> 11) & 0xF
        if k == K_JALR:
            imm = (w >> 15) - (1 << 17 if w >> 31 else 0)
            t = (int(rows0["registers"][i, fb]) + imm) & ((1 << 64) - 1)
            assert int(rows0["pc"][i + 1]) == t & ~1 and m[C_B0, i] == t & 1
            assert [int(m[C_Y + l, i]) for l in range(3)] == [(int(rows0["pc"][i]) + 4) & 0xFFFFF, (int(rows0["pc"][i]) + 4) >> 20 & 0xFFFFF, 0]
            if fa: assert int(rows0["registers"][i + 1, fa]) == int(rows0["pc"][i]) + 4 and m[C_WR + fa - 1, i] == 1
            else: assert not m[C_WR:C_WR + 15, i].any()
            seen.add(("jalr", int(m[C_B0, i]), imm < 0))
        if k == K_BRU:
            assert not m[C_WR:C_WR + 15, i].any() and not cls[K_OJ][i]
            seen.add((op, int(rows0["pc"][i + 1]) != int(rows0["pc"][i]) + 4))
        if k == K_OTH:
            assert int(rows0["pc"][i + 1]) == int(rows0["pc"][i]) + 4
            seen.add(op)
    assert seen >= {("jalr", 0, False), ("jalr", 1, True), (0x42, False), (0x42, True), (0x43, False), (0x43, True), 0x02, 0x1B}
    assert so.verify(so.prove(rows0, pub)) == 0

    def rejected(rows):
        return so.verify(so.prove(rows, pub)) == 10
    kj = int(np.nonzero(ops == 0x49)[0][3]); km = int(np.nonzero(ops == 0x02)[0][3]); kb = int(np.nonzero(ops == 0x42)[0][3])
    code_pcs = [0x1000 + 4 * t for t in range(len(spec.call_loop_program().code))]
    # a JALR that returns somewhere else (another valid code address): only that row is forged, the rows after it are the honest ones
    for other in (int(rows0["pc"][kj + 1]) + 4, int(rows0["pc"][kj + 1]) - 4, code_pcs[0]):
        rows = rows0.copy(); rows["pc"][kj + 1] = other
        assert rejected(rows), hex(other)
    # ... with a wrong link value
    rows = rows0.copy(); rd = (int(rows0["instruction"][kj]) >> 7) & 0xF
    kj13 = int(np.nonzero((ops == 0x49) & (((rows0["instruction"] >> 7) & 0xF) == 13))[0][0])
    rows["registers"][kj13 + 1:, 13] += 4
    assert rejected(rows)
    # an "other" row (MUL) that jumps: the next row sits at another code address
    rows = rows0.copy(); rows["pc"][km + 1] = int(rows0["pc"][km]) + 8
    assert rejected(rows)
    # a signed branch goes the way its comparison says (v5), and it cannot write a register
    taken = int(rows0["pc"][kb + 1]) != int(rows0["pc"][kb]) + 4
    wb = int(rows0["instruction"][kb]); immb = (wb >> 15) - (1 << 17 if wb >> 31 else 0)
    rows = rows0.copy(); rows["pc"][kb + 1] = int(rows0["pc"][kb]) + (4 if taken else immb)
    assert rejected(rows)
    rows = rows0.copy(); rows["registers"][kb + 1:, 9] = 77
    assert rejected(rows)
    m2 = m.copy(); m2[C_WR + 8, kb] = 1; m2[C_LIMB + 27, kb + 1:] = 77       # ... even if the matrix flags the write
    assert so.verify(so.prove_matrix(m2, pub)) == 10
    # a JALR row relabelled as the free-pc class: its word's class says otherwise (ROM tuple)
    m2 = m.copy(); m2[C_K3, kj] = 0; m2[C_K3 + 1, kj] = 1
    assert so.verify(so.prove_matrix(m2, pub)) == 10
    m2[C_OPC, kj] = K_OJ
    assert so.verify(so.prove_matrix(m2, pub)) == 10
    m2 = m.copy(); m2[C_K2 + 1, kb] = 0; m2[C_K3 + 1, kb] = 1; m2[C_B0, kb] = 0; m2[C_SB, kb] = 0; m2[C_TK, kb] = 0     # a signed branch relabelled as the free-pc class (deferred mode's)
    assert so.verify(so.prove_matrix(m2, pub)) == 10
    # the cleared bit forged together with the low pc limb: pc' stops being a code address
    m2 = m.copy(); m2[C_B0, kj] = 1 - int(m[C_B0, kj]); m2[C_PC, kj + 1] = (int(m[C_PC, kj + 1]) + (1 if m[C_B0, kj] else -1)) % P
    assert so.verify(so.prove_matrix(m2, pub)) == 10


def test_cheating_prover_matrices_are_rejected():
    """A prover that submits its own main-trace matrix: relabelling a constrained opcode as 'other', freeing the write selector,
    lying about an operand, or skipping rows by early padding."""
    rows, pub = _run(40)
    m0 = so.main_trace(rows, pub)
    assert so.verify(so.prove_matrix(m0, pub)) == 0
    ops = rows["instruction"] & 0x7F
    k = int(np.nonzero(ops == 0x00)[0][2])

    def bad(edit):
        m = m0.copy(); edit(m)
        return so.verify(so.prove_matrix(m, pub)) == 10

    def relabel(m):                                              # ADD row claims to be "other" to escape the addition constraint
        m[C_K + K_ADD, k] = 0; m[C_K + K_OTH, k] = 1
    assert bad(relabel)

    def relabel_with_opclass(m):                                 # ... and forges the word's class as well: the tuple is then not a ROM row
        relabel(m); m[C_OPC, k] = 4
    assert bad(relabel_with_opclass)

    def out_of_range_carry(m):                                   # VERDICT r2 missing #2: a carry the ranges forbid.  ADD row k: flip c0 and keep every
        nxt = k + 1 + int(np.nonzero(ops[k + 1:] == 0x00)[0][0])  # constraint of the addition satisfied: y0 shifts by 2^20 (out of range), its high chunk by 1024
        c0 = int(m[C_C0, k]); d = 1 - 2 * c0
        m[C_C0, k] = 1 - c0
        m[C_Y, k] = (int(m[C_Y, k]) - d * (1 << 20)) % P
        m[C_RC + 1, k] = (int(m[C_RC + 1, k]) - d * 1024) % P
        y1 = int(m[C_Y + 1, k]) + d                               # the carry into limb 1 changes with it; stays inside 20 bits here
        assert 0 <= y1 < 1 << 20
        m[C_Y + 1, k] = y1; m[C_RC + 2, k] = y1 & 1023; m[C_RC + 3, k] = y1 >> 10
        rd = 4                                                    # add r4, r1, r2: the forged limbs are what the register shows until it is written again
        m[C_LIMB + 3 * rd, k + 1:nxt + 1] = m[C_Y, k]; m[C_LIMB + 3 * rd + 1, k + 1:nxt + 1] = m[C_Y + 1, k]
    assert bad(out_of_range_carry)
    assert bad(lambda m: m.__setitem__((C_WR + 6, k), 1))        # a second written register
    assert bad(lambda m: m.__setitem__((C_XB, k), (int(m[C_XB, k]) + 1) % P))          # operand not the register selected by field b
    assert bad(lambda m: m.__setitem__((C_SELB + 2, k), 1))      # selector not one-hot

    def wrong_sum(m):                                            # y off by one, and the next rows consistently show the wrong value
        nxt = k + 1 + int(np.nonzero(ops[k + 1:] == 0x00)[0][0])
        m[C_Y, k] = (int(m[C_Y, k]) + 1) % P
        m[C_LIMB + 3 * 4, k + 1:nxt + 1] = (m[C_LIMB + 3 * 4, k + 1:nxt + 1].astype(np.int64) + 1) % P
    assert bad(wrong_sum)

    def wrong_sum_with_z(m):                                     # ... the same with z and its chunks moved along (y = z holds): the addition itself fails
        wrong_sum(m)
        m[C_RC, k] = int(m[C_Y, k]) & 1023; m[C_RC + 1, k] = int(m[C_Y, k]) >> 10
    assert bad(wrong_sum_with_z)

    # AIR v3: the same on the comparison families (rows of spec.compare_loop_program)
    rows3, pub3 = _run(600, "cmp")
    m3 = so.main_trace(rows3, pub3)
    ops3 = rows3["instruction"] & 0x7F

    def bad3(edit):
        m = m3.copy(); edit(m)
        return so.verify(so.prove_matrix(m, pub3)) == 10
    ku = int(np.nonzero(ops3 == 0x20)[0][5]); kb = int(np.nonzero(ops3 == 0x44)[0][5]); ke = int(np.nonzero(ops3 == 0x40)[0][5]); ks = int(np.nonzero(ops3 == 0x01)[0][5])

    def flip_borrow(m, k):                                       # the borrow that decides an unsigned comparison, flipped with z moved along by 2^20 x 2^20:
        c1 = int(m[C_C0 + 1, k]); d = 1 - 2 * c1                 # every difference constraint still holds, z1 leaves its range — the chunk lookup refuses it
        m[C_C0 + 1, k] = 1 - c1
        m[C_RC + 3, k] = (int(m[C_RC + 3, k]) + d * 1024) % P
        m[C_FLAG, k] = 1 - c1; m[C_FX, k] = 1 - int(m[C_FX, k])
    def sltu_flipped(m):                                          # ... with the written value (and what the register shows afterwards) following the forged flag
        flip_borrow(m, ku); m[C_Y, ku] = m[C_FX, ku]
        nxt = ku + 1 + int(np.nonzero(ops3[ku + 1:] == 0x20)[0][0])
        m[C_LIMB + 3 * 6, ku + 1:nxt + 1] = m[C_FX, ku]
    assert bad3(sltu_flipped)
    assert bad3(lambda m: (flip_borrow(m, kb), m.__setitem__((C_TK, kb), m[C_FX, kb])))          # BLTU decided by a forged borrow
    assert bad3(lambda m: m.__setitem__((C_FLAG, ke), 1 - int(m[C_FLAG, ke])))                    # the equality flag is not [xb == xc]
    assert bad3(lambda m: m.__setitem__((C_FX, ke), 1 - int(m[C_FX, ke])))                        # the polarity is not the opcode's
    assert bad3(lambda m: m.__setitem__((C_TK, ke), 1 - int(m[C_TK, ke])))                        # the branch does not follow fx
    assert bad3(lambda m: (m.__setitem__((C_K2 + 0, ks), 0), m.__setitem__((C_K + K_OTH, ks), 1)))  # a SUB row hiding as "other"
    assert bad3(lambda m: (m.__setitem__((C_K2 + 0, ks), 0), m.__setitem__((C_K + K_ADD, ks), 1)))  # ... or running as an ADD

    def early_pad(m):                                            # stop executing at row 20: halt there, padding afterwards
        m[C_K:C_K + 7, 20:] = 0; m[C_K + K_HALT, 20] = 1; m[C_K + K_PAD, 21:] = 1
        m[1:C_K, 21:] = m[1:C_K, 20:21]; m[C_WR:C_WR + 15, 20:] = 0; m[C_Y:C_Y + 3, 20:] = 0; m[C_C0:C_C0 + 5, 20:] = 0; m[C_TK, 20:] = 0
        m[C_RC:C_RC + 4, 20:] = 0; m[C_FLAG, 20:] = 0; m[C_FX, 20:] = m[C_OP, 20:]
    assert bad(early_pad)                                        # the public row count pins the halt row (is_last)


def test_transcript_binds_everything():
    a = so.prove(*_run(32))
    al, ze, ga = so.last_challenges()
    rows, pub = _run(32)
    rows = rows.copy()
    rows["pc"][7] ^= 4
    b = so.prove(rows, pub)
    al2, ze2, ga2 = so.last_challenges()
    assert not np.array_equal(al, al2) and not np.array_equal(ze, ze2) and not np.array_equal(ga, ga2)
    t0 = so.proof_layout(a)["trace_root"]
    assert not np.array_equal(a[t0:t0 + 4], b[t0:t0 + 4])        # trace roots differ
