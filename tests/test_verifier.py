"""The product's verifier (zkir_verify, zkir_amd/csrc/verify.cpp: Montgomery arithmetic + the air.h constraint template shared with
the quotient kernel) against the oracle's (so::verify, naive arithmetic, its own constraint list): same verdict — and the same
failing check — on valid proofs, on every kind of tampering and on cheating provers' proofs.  Host only: the proofs here are made
by the oracle prover, so this also pins air.h's constraint indices to the oracle's order (any mismatch fails check 10)."""
import numpy as np
import pytest

from oracle import api as oracle, stark_api as so
from zkir_amd import runtime as rt, spec

P = so.P


def _pub_c(p: so.PublicC) -> rt.PublicInputsC:
    out = rt.PublicInputsC(p.n_real, p.entry, p.deferred, 0)
    out.program_digest[:] = list(p.prog); out.io_digest[:] = list(p.io)
    return out


def _run(n, prog="fib", **cfg):
    blob = {"fib": spec.fib_endless_program, "sha": spec.sha256_chain_program, "fib12": lambda: spec.fib_program(12)}[prog]().to_bytes()
    res = oracle.run(blob, max_cycles=n or 1_000_000, enable_execution_trace=True, **cfg)
    return res.rows, so.public_inputs(len(res.rows), blob, [], list(res.outputs), (res.halt_kind, res.halt_code), deferred=bool(cfg))


@pytest.mark.parametrize("n,prog,cfg", [(8, "fib", {}), (5, "fib", {}), (100, "fib", {}), (300, "sha", {}), (None, "fib12", {}), (512, "fib", {}),
                                         (200, "fib", {"enable_deferred_model": True})])
def test_accepts_what_the_oracle_accepts(n, prog, cfg):
    rows, pub = _run(n, prog, **cfg)
    pr = so.prove(rows, pub)
    assert so.verify(pr, pub) == 0
    assert rt.verify(pr) == 0 and rt.verify(pr, _pub_c(pub)) == 0
    rng = np.random.default_rng(len(rows))
    for pos in list(range(2, 21)) + [int(x) for x in rng.integers(21, len(pr), 60)] + [len(pr) - 1]:     # same verdict AND same failing check everywhere
        t = pr.copy()
        t[pos] = (int(t[pos]) + 1 + int(rng.integers(0, 50))) % P
        if t[pos] != pr[pos]:
            want = so.verify(t)
            assert want != 0 and rt.verify(t) == want, (pos, want, rt.verify(t))
    assert rt.verify(pr[:-1]) == so.verify(pr[:-1]) != 0
    assert rt.verify(np.concatenate([pr, [0]])) == so.verify(np.concatenate([pr, [0]])) == 30
    t = pr.copy(); t[40] = P                                                                              # non-canonical word
    assert rt.verify(t) == so.verify(t) == 3


def test_public_inputs_and_digests():
    blob = spec.fib_program(12).to_bytes()
    for data in (b"", b"a", b"hello", blob, bytes(range(256)) * 5):
        got = np.zeros(4, np.uint32)
        rt.lib().zkir_digest_bytes(data, len(data), got.ctypes.data)
        assert np.array_equal(got, so.digest_bytes(data))
    log = rt.interpret(blob, [7, 8], rt.VMConfig(enable_execution_trace=True))
    pub = rt.public_inputs(log, blob, [7, 8])
    want = so.public_inputs(log.cycles, blob, [7, 8], log.outputs, (log.halt_reason.kind, log.halt_reason.code))
    assert (pub.n_real, pub.entry_point, pub.deferred) == (want.n_real, want.entry, 0)
    assert list(pub.program_digest) == list(want.prog) and list(pub.io_digest) == list(want.io)
    rows, opub = _run(None, "fib12")
    pr = so.prove(rows, opub)
    assert rt.verify(pr, _pub_c(opub)) == 0
    other = _pub_c(so.public_inputs(len(rows), blob, [], [999], (1, 0)))
    assert rt.verify(pr, other) == 6
    log.close()


def test_rejects_cheating_provers_like_the_oracle():
    rows, pub = _run(40)
    m0 = so.main_trace(rows, pub)
    ops = rows["instruction"] & 0x7F
    k = int(np.nonzero(ops == 0x00)[0][2])
    C_K, C_T, C_WR, C_XB, C_Y = 127, 134, 73, 118, 124

    def relabel(m):
        m[C_K + 0, k] = 0; m[C_K + 4, k] = 1; m[C_T + 4, k] = 1
    edits = [relabel, lambda m: m.__setitem__((C_WR + 6, k), 1), lambda m: m.__setitem__((C_XB, k), (int(m[C_XB, k]) + 1) % P),
             lambda m: m.__setitem__((C_Y, k), (int(m[C_Y, k]) + 1) % P)]
    for e in edits:
        m = m0.copy(); e(m)
        pr = so.prove_matrix(m, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10
    for mutate in (lambda r: r["registers"].__setitem__((slice(k + 1, k + 4), 4), 5), lambda r: r["pc"].__setitem__(k + 1, 0x2000)):
        r = rows.copy(); mutate(r)
        pr = so.prove(r, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10
