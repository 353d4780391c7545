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
    blob = {"fib": spec.fib_endless_program, "sha": spec.sha256_chain_program, "fib12": lambda: spec.fib_program(12), "cmp": spec.compare_loop_program, "call": spec.call_loop_program, "sgn": spec.signed_loop_program, "cmov": spec.cmov_loop_program}[prog]().to_bytes()
    res = oracle.run(blob, max_cycles=n or 1_000_000, enable_execution_trace=True, **cfg)
    return res.rows, so.public_inputs(len(res.rows), blob, [], list(res.outputs), (res.halt_kind, res.halt_code), deferred=bool(cfg))


@pytest.mark.parametrize("n,prog,cfg", [(8, "fib", {}), (5, "fib", {}), (100, "fib", {}), (300, "sha", {}), (None, "fib12", {}), (512, "fib", {}),
                                         (200, "fib", {"enable_deferred_model": True}), (600, "cmp", {}), (150, "cmp", {"enable_deferred_model": True}), (500, "call", {}),
                                         (200, "call", {"enable_deferred_model": True}), (700, "sgn", {}), (150, "sgn", {"enable_deferred_model": True}), (400, "cmov", {}), (150, "cmov", {"enable_deferred_model": True})])
def test_accepts_what_the_oracle_accepts(n, prog, cfg):
    rows, pub = _run(n, prog, **cfg)
    pr = so.prove(rows, pub)
    assert so.verify(pr, pub) == 0
    assert rt.verify(pr) == 0 and rt.verify(pr, _pub_c(pub)) == 0
    rng = np.random.default_rng(len(rows))
    for pos in list(range(2, 21)) + [21, 24, 30, 88, 89, 92, 156] + [int(x) for x in rng.integers(157, len(pr), 60)] + [len(pr) - 1]:     # same verdict AND same failing check everywhere
        t = pr.copy()
        t[pos] = (int(t[pos]) + 1 + int(rng.integers(0, 50))) % P
        if t[pos] != pr[pos]:
            want = so.verify(t)
            assert want != 0 and rt.verify(t) == want, (pos, want, rt.verify(t))
    assert rt.verify(pr[:-1]) == so.verify(pr[:-1]) != 0
    assert rt.verify(np.concatenate([pr, [0]])) == so.verify(np.concatenate([pr, [0]])) == 30
    t = pr.copy(); t[40] = P                                                                              # non-canonical word
    assert rt.verify(t) == so.verify(t) == 3


@pytest.mark.parametrize("seed", range(6))
def test_honest_runs_of_random_programs_are_accepted(seed):
    """Completeness of the AIR: an honest execution of ANY program (every opcode class, data-dependent branches, JALR, syscalls, either
    mode) that stays inside its code segment satisfies every constraint — class "other" rows really advance the pc by 4 and write at
    most one register, B-type rows write nothing.  Oracle prover, both verifiers."""
    import programs
    blob, inputs = programs.random_program(1000 + seed, n_instr=150)
    deferred = bool(seed & 1)
    try:
        res = oracle.run(blob, inputs, max_cycles=200 + 37 * seed, enable_execution_trace=True, enable_deferred_model=deferred)
    except oracle.OracleError:
        pytest.skip("this random program errors out (no trace to prove)")
    pub = so.public_inputs(len(res.rows), blob, list(inputs), list(res.outputs), (res.halt_kind, res.halt_code), deferred=deferred)
    pr = so.prove(res.rows, pub)
    code = so.verify(pr, pub)
    if code != 0:                                                 # only acceptable reason: the run left its code segment (no ROM row for that pc)
        in_code = (res.rows["pc"] >= 0x1000) & (res.rows["pc"] < 0x1000 + int.from_bytes(blob[16:20], "little"))
        assert not in_code.all(), code
        pytest.skip("this random program jumps out of its code segment: unprovable by design")
    assert rt.verify(pr) == 0 and rt.verify(pr, _pub_c(pub)) == 0


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
    C_K, C_OPC, C_RC, C_WR, C_XB, C_Y, C_FA = 127, 134, 135, 73, 118, 124, 5

    def relabel(m):                                               # an ADD row runs as "other", with the word's class forged to match: not a ROM row
        m[C_K + 0, k] = 0; m[C_K + 4, k] = 1; m[C_OPC, k] = 4

    def chunk_out_of_table(m):                                    # y0 = chunk0 + 1024 chunk1 still holds, but chunk1 is not a table entry
        m[C_RC, k] = (int(m[C_RC, k]) + 1024) % P; m[C_RC + 1, k] = (int(m[C_RC + 1, k]) - 1) % P
    edits = [relabel, lambda m: m.__setitem__((C_WR + 6, k), 1), lambda m: m.__setitem__((C_XB, k), (int(m[C_XB, k]) + 1) % P),
             lambda m: m.__setitem__((C_Y, k), (int(m[C_Y, k]) + 1) % P), chunk_out_of_table,
             lambda m: m.__setitem__((C_FA, k), int(m[C_FA, k]) ^ 1)]                     # another rd than the program's word has
    for e in edits:
        m = m0.copy(); e(m)
        pr = so.prove_matrix(m, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10
    for mutate in (lambda r: r["registers"].__setitem__((slice(k + 1, k + 4), 4), 5), lambda r: r["pc"].__setitem__(k + 1, 0x2000),
                   lambda r: r["instruction"].__setitem__(k, (int(r["instruction"][k]) & ~0x7F) | 0x01)):      # ADD -> SUB: the forged word AIR v1 accepted
        r = rows.copy(); mutate(r)
        pr = so.prove(r, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10


def test_rejects_forged_opcode_families_like_the_oracle():
    """AIR v3 (SUB, SLTU / SGEU, SEQ / SNE, BEQ / BNE, BLTU / BGEU): the product verifier's constraint list (air.h) and the oracle's give
    the same verdict on forged comparison flags, polarities, branch decisions, differences, and on rows relabelled into another class."""
    rows, pub = _run(600, "cmp")
    m0 = so.main_trace(rows, pub)
    ops = rows["instruction"] & 0x7F
    C_K, C_K2, C_Y, C_C1, C_TK, C_RC, C_FLAG, C_FX = 127, 152, 124, 142, 151, 135, 158, 159
    at = {op: int(np.nonzero(ops == op)[0][4]) for op in (0x01, 0x20, 0x21, 0x24, 0x25, 0x40, 0x41, 0x44, 0x45)}
    flip = lambda col, k: (lambda m: m.__setitem__((col, k), 1 - int(m[col, k])))
    edits = [flip(C_FLAG, at[0x24]), flip(C_FX, at[0x25]), flip(C_TK, at[0x40]), flip(C_TK, at[0x45]), flip(C_FLAG, at[0x44]), flip(C_C1, at[0x20]), flip(C_C1, at[0x44]),
             flip(C_Y, at[0x21]), lambda m: m.__setitem__((C_RC, at[0x01]), (int(m[C_RC, at[0x01]]) + 1) % 1024),
             lambda m: (m.__setitem__((C_K2 + 0, at[0x01]), 0), m.__setitem__((C_K + 4, at[0x01]), 1)),          # SUB as "other"
             lambda m: (m.__setitem__((C_K2 + 3, at[0x20]), 0), m.__setitem__((C_K2 + 2, at[0x20]), 1)),         # SLTU as the equality family
             lambda m: (m.__setitem__((C_K2 + 1, at[0x44]), 0), m.__setitem__((C_K + 2, at[0x44]), 1))]          # BLTU as BEQ / BNE
    for i, e in enumerate(edits):
        m = m0.copy(); e(m)
        pr = so.prove_matrix(m, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10, i
    pr = so.prove_matrix(m0, pub)
    assert so.verify(pr) == 0 and rt.verify(pr) == 0


def test_rejects_bent_control_flow_like_the_oracle():
    """AIR v4 / v5 (JALR, sequential "other" rows, BLT / BGE): same verdict from the product verifier and the oracle on
    forged jump targets, links, cleared bits, relabelled rows and writes by a signed branch."""
    rows, pub = _run(500, "call")
    m0 = so.main_trace(rows, pub)
    ops = rows["instruction"] & 0x7F
    C_PC, C_LIMB, C_WR, C_Y, C_K, C_OPC, C_D0, C_K3, C_B0 = 1, 9, 73, 124, 127, 134, 143, 160, 162
    kj, km, kb = (int(np.nonzero(ops == op)[0][3]) for op in (0x49, 0x02, 0x42))
    edits = [lambda m: m.__setitem__((C_B0, kj), 1 - int(m[C_B0, kj])), lambda m: m.__setitem__((C_D0, kj), 1 - int(m[C_D0, kj])),
             lambda m: m.__setitem__((C_Y, kj), (int(m[C_Y, kj]) + 4) % P),
             lambda m: (m.__setitem__((C_K3, kj), 0), m.__setitem__((C_K3 + 1, kj), 1)),                   # JALR as the free-pc class
             lambda m: (m.__setitem__((C_K3, kj), 0), m.__setitem__((C_K3 + 1, kj), 1), m.__setitem__((C_OPC, kj), 12)),
             lambda m: (m.__setitem__((C_K + 4, km), 0), m.__setitem__((C_K3 + 1, km), 1)),                 # MUL as the free-pc class
             lambda m: m.__setitem__((C_WR + 8, kb), 1)]                                                     # BLT flags a write
    for i, e in enumerate(edits):
        m = m0.copy(); e(m)
        pr = so.prove_matrix(m, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10, i
    for mutate in (lambda r: r["pc"].__setitem__(kj + 1, int(rows["pc"][kj + 1]) + 4), lambda r: r["pc"].__setitem__(km + 1, int(rows["pc"][km]) + 8),
                   lambda r: r["registers"].__setitem__((slice(kb + 1, None), 9), 5)):
        r = rows.copy(); mutate(r)
        pr = so.prove(r, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10
    pr = so.prove(rows, pub)
    assert so.verify(pr) == 0 and rt.verify(pr) == 0


def test_rejects_forged_signed_comparisons_like_the_oracle():
    """AIR v5 (SLT / SGE / BLT / BGE: variant bit, sign bits, biased high limbs as the second range-checked pair): same verdict from the
    product verifier and the oracle on forged sign bits, variant bits, borrows, flags and branch decisions, on signed comparisons written
    the wrong way round and signed branches going the other way."""
    rows, pub = _run(700, "sgn")
    m0 = so.main_trace(rows, pub)
    ops = rows["instruction"] & 0x7F
    C_C1, C_TK, C_FLAG, C_FX, C_B0, C_RC2, C_G, C_SB = 142, 151, 158, 159, 162, 163, 167, 168
    ksl = next(int(k) for k in np.nonzero(ops == 0x22)[0] if m0[C_B0, k] != m0[C_SB, k])
    kbg = next(int(k) for k in np.nonzero(ops == 0x43)[0] if m0[C_B0, k] != m0[C_SB, k])
    flip = lambda col, k: (lambda m: m.__setitem__((col, k), 1 - int(m[col, k])))

    def flip_sign(col, rc, k):                                    # the sign bit flipped with the biased limb following it: a chunk leaves the table
        def e(m):
            sgn = int(m[col, k]); m[col, k] = 1 - sgn
            m[rc + 1, k] = (int(m[rc + 1, k]) + (1024 if sgn else -1024)) % P
        return e
    edits = [flip(C_B0, ksl), flip(C_SB, ksl), flip(C_B0, kbg), flip(C_SB, kbg), flip_sign(C_B0, C_RC2, ksl), flip_sign(C_SB, C_RC2 + 2, kbg),
             flip(C_G, ksl), flip(C_G, kbg), flip(C_C1, ksl), flip(C_FLAG, ksl), flip(C_FX, kbg), flip(C_TK, kbg)]
    for i, e in enumerate(edits):
        m = m0.copy(); e(m)
        pr = so.prove_matrix(m, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10, i
    rd = (int(rows["instruction"][ksl]) >> 7) & 0xF
    wb = int(rows["instruction"][kbg]); imm = (wb >> 15) - (1 << 17 if wb >> 31 else 0)
    taken = int(rows["pc"][kbg + 1]) != int(rows["pc"][kbg]) + 4
    for mutate in (lambda r: r["registers"].__setitem__((ksl + 1, rd), int(rows["registers"][ksl + 1, rd]) ^ 1),
                   lambda r: r["pc"].__setitem__(kbg + 1, int(rows["pc"][kbg]) + (4 if taken else imm))):
        r = rows.copy(); mutate(r)
        pr = so.prove(r, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10
    pr = so.prove(rows, pub)
    assert so.verify(pr) == 0 and rt.verify(pr) == 0


def test_rejects_forged_conditional_moves_like_the_oracle():
    """AIR v6 (CMOV / CMOVNZ / CMOVZ; the range check of the bits above 40 an "other" row writes): same verdict from the product verifier and
    the oracle on forged conditions, write selectors, classes, moved values and an out-of-range third limb."""
    rows, pub = _run(400, "cmov")
    m0 = so.main_trace(rows, pub)
    ops = rows["instruction"] & 0x7F
    C_LIMB, C_WR, C_Y, C_K, C_NZ, C_IVZ, C_RC2, C_K4, C_Q = 9, 73, 124, 127, 156, 157, 163, 169, 171
    kq = next(int(k) for k in np.nonzero((ops == 0x26) & (((rows["instruction"] >> 7) & 0xF) == 11))[0] if m0[C_Q, k]); kn = next(int(k) for k in np.nonzero(ops == 0x27)[0] if not m0[C_Q, k])
    kl = int(np.nonzero(ops == 0x30)[0][0])
    flip = lambda col, k: (lambda m: m.__setitem__((col, k), 1 - int(m[col, k])))

    def y2_out_of_range(m):
        m[C_Y + 2, kl] = (int(m[C_Y + 2, kl]) + (1 << 24)) % P; m[C_RC2 + 2, kl] = int(m[C_RC2 + 2, kl]) + 16; m[C_RC2 + 3, kl] = 64 * int(m[C_RC2 + 2, kl])
        m[C_LIMB + 3 * 7 + 2, kl + 1:] = m[C_Y + 2, kl]
    edits = [flip(C_Q, kq), flip(C_Q, kn), flip(C_NZ, kq), flip(C_NZ, kn), flip(C_WR + 10, kq), lambda m: m.__setitem__((C_WR + 11, kn), 1),
             lambda m: (m.__setitem__((C_K4, kq), 0), m.__setitem__((C_K4 + 1, kq), 1)), lambda m: (m.__setitem__((C_K4, kq), 0), m.__setitem__((C_K + 4, kq), 1)),
             lambda m: m.__setitem__((C_Y, kq), (int(m[C_Y, kq]) + 1) % P), lambda m: m.__setitem__((C_IVZ, kq), (int(m[C_IVZ, kq]) + 1) % P), y2_out_of_range]
    for i, e in enumerate(edits):
        m = m0.copy(); e(m)
        pr = so.prove_matrix(m, pub)
        assert so.verify(pr) == 10 and rt.verify(pr) == 10, i
    rd = (int(rows["instruction"][kq]) >> 7) & 0xF
    r = rows.copy(); r["registers"][kq + 1, rd] = rows["registers"][kq, rd]              # the move undone
    pr = so.prove(r, pub)
    assert so.verify(pr) == 10 and rt.verify(pr) == 10
    pr = so.prove(rows, pub)
    assert so.verify(pr) == 0 and rt.verify(pr) == 0


@pytest.mark.parametrize("n_total,seg,prog", [(100, 40, "fib"), (300, 128, "sha"), (65, 33, "fib")])
def test_segment_chains_same_verdicts_as_the_oracle(n_total, seg, prog):
    """zkir_verify_segment / zkir_verify_chain against so::verify / so::verify_chain on a run proven in overlapping segments: accepted
    chains, and the same code for every way of breaking one (wrong start, gap, order, public inputs, row count, a bad segment)."""
    from test_stark_oracle import _segments
    rows, full, cuts, proofs = _segments(n_total, seg, prog)
    fullc = _pub_c(full)
    assert so.verify_chain(proofs, full) == 0 and rt.verify_chain(proofs, fullc) == 0 and rt.verify_chain(proofs) == 0
    for i, pr in enumerate(proofs):
        rc, f, l = rt.verify_segment(pr)
        orc, of, ol = so.verify_segment(pr)
        assert rc == orc == 0 and np.array_equal(f, of) and np.array_equal(l, ol) and int(f[0]) == cuts[i][0] and int(l[0]) == cuts[i][1] - 1
        assert rt.verify(pr) == so.verify(pr) == (0 if i == 0 else 7)
    wrong_rows = rt.PublicInputsC.from_buffer_copy(fullc); wrong_rows.n_real += 1
    wrong_prog = rt.PublicInputsC.from_buffer_copy(fullc); wrong_prog.program_digest[1] ^= 1
    o_rows = so.PublicC.from_buffer_copy(full); o_rows.n_real += 1
    o_prog = so.PublicC.from_buffer_copy(full); o_prog.prog[1] ^= 1
    tampered = proofs[-1].copy(); tampered[300] = (int(tampered[300]) + 1) % P
    cases = [(proofs[1:], None, None), (proofs[::-1], None, None), ([proofs[0], proofs[0]], None, None), (proofs, wrong_rows, o_rows), (proofs, wrong_prog, o_prog),
             (proofs[:-1] + [tampered], None, None), ([], None, None)]
    for chain, e_rt, e_so in cases:
        want = so.verify_chain(chain, e_so) if chain else 40
        assert want != 0 and rt.verify_chain(chain, e_rt) == want, (want, rt.verify_chain(chain, e_rt))


def test_verifier_fuzz_never_accepts_and_agrees_with_the_oracle():
    """1500 random corruptions of a valid proof — single words set to arbitrary 32-bit values (header fields included: sizes, counts,
    log2 N up to absurd values), random truncations and extensions, swapped regions: the product's verifier never accepts, never
    reads out of bounds (it would crash here), and names the same failing check as the oracle's."""
    rows, pub = _run(200)
    pr = so.prove(rows, pub)
    rng = np.random.default_rng(2024)
    special = [0, 1, 2, 3, 7, 8, 26, 27, 31, 32, 50, 64, 152, 153, 255, 256, 1 << 16, (1 << 20) - 1, 1 << 20, (1 << 30) - 1, 1 << 30, P - 1, P, P + 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF]
    for it in range(1500):
        t = pr.copy()
        kind = it % 5
        if kind == 0:                                             # a header / parameter word
            t[int(rng.integers(0, 165))] = special[int(rng.integers(0, len(special)))]
        elif kind == 1:                                           # any word, arbitrary value
            t[int(rng.integers(0, len(t)))] = int(rng.integers(0, 1 << 32))
        elif kind == 2:                                           # truncation / extension
            cut = int(rng.integers(0, len(t) + 40))
            t = t[:cut] if cut <= len(t) else np.concatenate([t, rng.integers(0, P, cut - len(t)).astype(np.uint32)])
        elif kind == 3:                                           # two words swapped
            a, b = (int(x) for x in rng.integers(0, len(t), 2))
            t[a], t[b] = t[b], t[a]
        else:                                                     # a run of words zeroed
            a = int(rng.integers(0, len(t)))
            t[a:a + int(rng.integers(1, 64))] = 0
        if len(t) == len(pr) and np.array_equal(t, pr):
            continue
        got, want = rt.verify(t), so.verify(t)
        assert got != 0 and got == want, (it, kind, got, want)


def test_chain_verifier_fuzz_agrees_with_the_oracle():
    """Random damage to a chain of segment proofs — a corrupted word in one segment, segments dropped, repeated or reordered, a
    truncated segment: zkir_verify_chain never accepts and returns the oracle's code."""
    from test_stark_oracle import _segments
    rows, full, cuts, proofs = _segments(200, 60)
    fullc = _pub_c(full)
    assert len(proofs) == 4 and rt.verify_chain(proofs, fullc) == 0
    rng = np.random.default_rng(7)
    for it in range(300):
        chain = [p.copy() for p in proofs]
        kind = it % 4
        if kind == 0:
            k = int(rng.integers(0, len(chain))); pos = int(rng.integers(0, len(chain[k])))
            chain[k][pos] = int(rng.integers(0, 1 << 32))
        elif kind == 1:
            order = rng.permutation(len(chain))
            if (order == np.arange(len(chain))).all():
                continue
            chain = [chain[i] for i in order]
        elif kind == 2:
            k = int(rng.integers(0, len(chain)))
            chain = chain[:k] + chain[k + 1:] if rng.integers(0, 2) else chain[:k] + [chain[k]] + chain[k:]
        else:
            k = int(rng.integers(0, len(chain)))
            chain[k] = chain[k][:int(rng.integers(0, len(chain[k])))]
        same = len(chain) == len(proofs) and all(len(a) == len(b) and np.array_equal(a, b) for a, b in zip(chain, proofs))
        if same:
            continue
        got, want = rt.verify_chain(chain, fullc), (so.verify_chain(chain, full) if chain else 40)
        assert got != 0 and got == want, (it, kind, got, want)


# ---- the claim in the clear: I/O tapes + halt reason (zkir_verify_io / so::verify_io) --------------------------------------------------------------
def _both_io(pr, pub, inputs, outputs, halt):
    a = so.verify_io(pr, pub, inputs, outputs, halt)
    b = rt.verify_io(pr, _pub_c(pub) if pub is not None else None, inputs, outputs, halt)
    assert a == b, (a, b)
    return a


@pytest.mark.parametrize("deferred", [False, True])
def test_verify_io_accepts_the_honest_claim_and_rejects_every_other(deferred):
    """fib(12) exits through ECALL with R10 = 0, R11 = 0 after writing one output (syscall.rs:101-121)."""
    blob = spec.fib_program(12).to_bytes()
    res = oracle.run(blob, enable_execution_trace=True, enable_deferred_model=deferred)
    halt = (res.halt_kind, res.halt_code)
    assert halt == (1, 0) and list(res.outputs) == [144]
    pub = so.public_inputs(len(res.rows), blob, [], list(res.outputs), halt, deferred=deferred)
    pr = so.prove(res.rows, pub)
    assert _both_io(pr, pub, [], [144], halt) == 0
    assert _both_io(pr, None, [], [144], halt) == 0
    assert _both_io(pr, pub, [], [145], halt) == 50            # another output: not what the digest was made of
    assert _both_io(pr, pub, [7], [144], halt) == 50           # another input tape
    assert _both_io(pr, pub, [], [144], (1, 1)) == 50          # another exit code
    assert _both_io(pr, pub, [], [144], (0, 0)) == 50          # another halt reason
    t = pr.copy(); t[30] = (int(t[30]) + 1) % P
    assert _both_io(t, pub, [], [144], halt) == so.verify(t, pub) != 0     # a proof that does not verify fails with the verifier's own code


def test_verify_io_rejects_a_run_cut_short_and_called_an_exit():
    """The cheat ADVICE r3 describes: the AIR lets ANY row be the last one, so a prover stops fib(12) ten rows early (on an ADDI), hashes the claim 'exited with
    code 0, output 144' into the io digest and proves that trace.  zkir_verify accepts it — the claim is only bound, not checked — zkir_verify_io looks the halt row's
    instruction up in the program: 52."""
    blob = spec.fib_program(12).to_bytes()
    res = oracle.run(blob, enable_execution_trace=True)
    k = len(res.rows) - 10
    claim = dict(inputs=[], outputs=[144], halt=(1, 0))
    pub = so.public_inputs(k, blob, claim["inputs"], claim["outputs"], claim["halt"])
    pr = so.prove(res.rows[:k], pub)
    assert so.verify(pr, pub) == 0 and rt.verify(pr, _pub_c(pub)) == 0
    assert _both_io(pr, pub, **claim) == 52
    # stopped ON the final ECALL's row but with another exit code claimed (R11 = 0 there): 53
    pub2 = so.public_inputs(len(res.rows), blob, [], [144], (1, 5))
    pr2 = so.prove(res.rows, pub2)
    assert so.verify(pr2, pub2) == 0
    assert _both_io(pr2, pub2, [], [144], (1, 5)) == 53
    # stopped on the WRITE ecall (R10 = 2) and called an exit: an ECALL, but not SYSCALL_EXIT: 53
    words = np.frombuffer(blob[32:32 + int.from_bytes(blob[16:20], "little")], dtype="<u4")
    ecall_rows = [i for i, r in enumerate(res.rows) if (int(r["instruction"]) & 0x7F) == 0x50]
    assert len(ecall_rows) == 2 and words.size
    k3 = ecall_rows[0] + 1
    pub3 = so.public_inputs(k3, blob, [], [], (1, 144))         # R11 = 144 on that row: even the 'exit code' matches, the syscall number does not
    pr3 = so.prove(res.rows[:k3], pub3)
    assert so.verify(pr3, pub3) == 0 and _both_io(pr3, pub3, [], [], (1, 144)) == 53


def test_verify_io_ebreak_and_cycle_limit():
    prog = spec.Program.from_code([spec.addi(1, 0, 5), spec.addi(2, 1, 7), spec.ebreak()])
    blob = prog.to_bytes()
    res = oracle.run(blob, enable_execution_trace=True)
    assert (res.halt_kind, len(res.rows)) == (0, 3)
    pub = so.public_inputs(3, blob, [], [], (0, 0))
    pr = so.prove(res.rows, pub)
    assert _both_io(pr, pub, [], [], (0, 0)) == 0
    pub_cut = so.public_inputs(2, blob, [], [], (0, 0))         # cut before the EBREAK, still called an Ebreak halt
    pr_cut = so.prove(res.rows[:2], pub_cut)
    assert so.verify(pr_cut, pub_cut) == 0 and _both_io(pr_cut, pub_cut, [], [], (0, 0)) == 52
    # a CycleLimit halt names no instruction: the claim is the cycle count (vm.rs:211-214)
    blob2 = spec.fib_endless_program().to_bytes()
    res2 = oracle.run(blob2, max_cycles=100, enable_execution_trace=True)
    pub2 = so.public_inputs(100, blob2, [], [], (2, 0))
    pr2 = so.prove(res2.rows, pub2)
    assert _both_io(pr2, pub2, [], [], (2, 0)) == 0
    assert _both_io(pr2, pub2, [], [], (0, 0)) == 50


def test_verify_chain_io():
    """A run proven in two segments: the claim is checked against the LAST segment's last state."""
    blob = spec.fib_program(12).to_bytes()
    res = oracle.run(blob, enable_execution_trace=True)
    n = len(res.rows)
    halt = (res.halt_kind, res.halt_code)
    run_pub = so.public_inputs(n, blob, [], list(res.outputs), halt)
    cut = n // 2
    segs = []
    for lo, hi in ((0, cut + 1), (cut, n)):
        p = so.public_inputs(hi - lo, blob, [], list(res.outputs), halt)
        p.io[:] = list(run_pub.io)
        segs.append(so.prove(res.rows[lo:hi], p))
    assert so.verify_chain(segs, run_pub) == 0
    assert so.verify_chain_io(segs, run_pub, [], [144], halt) == rt.verify_chain_io(segs, _pub_c(run_pub), [], [144], halt) == 0
    assert so.verify_chain_io(segs, run_pub, [], [143], halt) == rt.verify_chain_io(segs, _pub_c(run_pub), [], [143], halt) == 50


def test_prover_parameters_are_part_of_the_statement():
    """zkir_prover_params (round 5): FRI queries and grinding bits are the PROOF's (header words 4 and 6, observed by the transcript).  A verifier with an `expect`
    requires exactly the expected ones (0 = the defaults, 50 + 12); without one, anything from the defaults up (never fewer).  Oracle and product verifier agree."""
    from zkir_amd import stark
    blob = spec.fib_program(30).to_bytes()
    ores = oracle.run(blob, [], enable_execution_trace=True)
    args = (len(ores.rows), blob, [], list(ores.outputs), (ores.halt_kind, ores.halt_code))
    dflt, big = so.public_inputs(*args), so.public_inputs(*args, num_queries=84, pow_bits=16)
    p0, p1 = so.prove(ores.rows, dflt), so.prove(ores.rows, big)
    assert (p0[4], p0[6], p1[4], p1[6]) == (50, 12, 84, 16) and len(p1) > len(p0)
    lay = stark.proof_layout(p1)
    assert (lay["mode"], lay["num_queries"], lay["pow_bits"]) == (0, 84, 16)

    def c_pub(p, **params):
        out = rt.PublicInputsC(p.n_real, p.entry, p.deferred, 0)
        out.program_digest[:] = list(p.prog); out.io_digest[:] = list(p.io)
        return out.with_params(**params) if params else out
    for proof, expect_o, expect_c, verdict in ((p0, dflt, c_pub(dflt), 0), (p1, big, c_pub(big, num_queries=84, pow_bits=16), 0), (p1, dflt, c_pub(dflt), 2), (p0, big, c_pub(big, num_queries=84, pow_bits=16), 2),
                                               (p0, None, None, 0), (p1, None, None, 0)):
        assert so.verify(proof, expect_o) == verdict and rt.verify(proof, expect_c) == verdict
    weak = p0.copy(); weak[4] = 49                                        # fewer queries than the defaults: refused whatever else the proof says
    assert so.verify(weak) == 2 and rt.verify(weak) == 2
    t = p1.copy(); t[6] = 17                                              # the parameters are observed: another claim about them is another transcript
    assert so.verify(t) != 0 and rt.verify(t) == so.verify(t)
    for bad in (dict(num_queries=49), dict(num_queries=129), dict(pow_bits=11), dict(pow_bits=25)):
        with pytest.raises(rt.RuntimeError) as e:
            c_pub(dflt, **bad)
        assert e.value.code == rt.ERR_ARGUMENT
