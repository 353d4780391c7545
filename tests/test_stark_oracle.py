"""Algebraic self-checks of the self-defined prover-stage oracle (oracle/stark_oracle.cpp).  These stages have no
reference implementation (parity unpinned), so the oracle is pinned by independent python-int
arithmetic here: field laws, naive DFT vs NTT, inverse round trips, LDE agreement on the original domain, Poseidon2
structural properties, Merkle recomputation."""
import numpy as np
import pytest

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


def test_main_trace_columns_and_commit():
    blob = spec.fib_endless_program().to_bytes()
    rows = oracle.run(blob, max_cycles=64, enable_execution_trace=True).rows
    m = so.main_trace(rows)
    assert m.shape == (89, 64) and (m < P).all()
    assert list(m[0]) == list(range(64))
    assert np.array_equal(m[1], rows["pc"] & 0xFFFFF) and not m[3].any()
    assert np.array_equal(m[4], rows["instruction"] & 0x7F)
    assert np.array_equal(m[9 + 3 * 4], rows["registers"][:, 4] & 0xFFFFF)
    # changed flags: content of the register triple differs in the next row
    ch4 = m[73 + 4]
    nxt = (rows["registers"][1:, 4] != rows["registers"][:-1, 4]) | (rows["bound_bits"][1:, 4] != rows["bound_bits"][:-1, 4])
    assert np.array_equal(ch4[:-1].astype(bool), nxt) and ch4[-1] == 0
    root, L = so.commit_trace(rows, 1, want_lde=True)
    assert L.shape == (89, 128)
    assert np.array_equal(so.merkle(L), root)
    coeffs, col = so.lde(m[0], 1)
    assert np.array_equal(col, L[0])


# ---- stage B: prover + verifier ------------------------------------------------------------------------------------
def _rows(n, prog="fib", **cfg):
    blob = (spec.fib_endless_program() if prog == "fib" else spec.sha256_chain_program()).to_bytes()
    return oracle.run(blob, max_cycles=n, enable_execution_trace=True, **cfg).rows


def _fri_schedule(log_n, log_final=3, log_arity=3):
    """Folds per committed FRI layer (oracle/stark_oracle.cpp: fri_schedule): layer 0 once, then 8-to-1, shorter at the end."""
    ks, log_m = [], log_n + 1
    while log_m > log_final:
        k = 1 if not ks else min(log_arity, log_m - log_final)
        ks.append(k)
        log_m -= k
    return ks


@pytest.mark.parametrize("log_n,prog", [(3, "fib"), (4, "fib"), (5, "fib"), (7, "fib"), (8, "fib"), (9, "sha"), (10, "fib")])
def test_prove_verify_roundtrip(log_n, prog):
    pr = so.prove(_rows(1 << log_n, prog))
    ks = _fri_schedule(log_n)                                                              # [1], [1,1], [1,2], [1,3,1], [1,3,2], [1,3,3], [1,3,3,1]
    assert pr[1] == 2 and pr[6 + 8 + (2 * 89 + 4) * 4] == len(ks)                          # proof version, number of committed FRI layers
    depth = [log_n + 1 - sum(ks[:j + 1]) for j in range(len(ks))]                          # Merkle depth of each FRI tree
    per_query = 1 + 2 * (89 + 4 * (log_n + 1)) + 2 * (4 + 4 * (log_n + 1)) + sum(4 * (1 << k) + 4 * d for k, d in zip(ks, depth))
    assert len(pr) == 6 + 8 + (2 * 89 + 4) * 4 + 1 + 4 * len(ks) + 4 * 8 + 24 * per_query
    assert pr[0] == 0x46504B5A and pr[2] == log_n and pr[3] == 89 and pr[4] == 24
    assert (pr[6:] < P).all()
    assert so.verify(pr) == 0
    assert so.verify(pr[:-1]) != 0 and so.verify(np.concatenate([pr, [0]])) != 0           # length is checked
    rng = np.random.default_rng(log_n)
    for pos in rng.integers(6, len(pr), 40):                                               # any single-word change is rejected
        t = pr.copy()
        t[pos] = (int(t[pos]) + 1 + int(rng.integers(0, 1000))) % P
        if t[pos] != pr[pos]:
            assert so.verify(t) != 0, pos


def test_quotient_is_low_degree_and_fri_layers_fold():
    n = 64
    so.prove(_rows(n))
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
    rows = _rows(64).copy()
    rows["cycle"][30] = 77                                   # cycle counter must increase by one
    assert so.verify(so.prove(rows)) == 10
    rows = _rows(64).copy()
    rows["registers"][:, 0] = 1                              # R0 is hard-wired zero
    assert so.verify(so.prove(rows)) == 10
    rows = _rows(64).copy()
    rows["reg_state"][10, 3] = 2                             # storage state is boolean
    assert so.verify(so.prove(rows)) == 10
    rows = _rows(64).copy()
    rows["cycle"] += 5                                       # first row must be cycle 0
    assert so.verify(so.prove(rows)) == 10


def test_transcript_binds_everything():
    a = so.prove(_rows(32))
    al, ze, ga = so.last_challenges()
    rows = _rows(32).copy()
    rows["pc"][7] ^= 4
    b = so.prove(rows)
    al2, ze2, ga2 = so.last_challenges()
    assert not np.array_equal(al, al2) and not np.array_equal(ze, ze2) and not np.array_equal(ga, ga2)
    assert not np.array_equal(a[6:10], b[6:10])              # trace roots differ
