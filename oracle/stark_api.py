"""ctypes view of oracle/stark_oracle.cpp (self-defined prover stages; PARITY UNPINNED).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api

P = 2013265921
W_MAIN = 172                # LOGICAL main-trace columns (what main_trace() returns and the constraints read)
W_COMMITTED = 152           # columns of the committed matrix in the default VM mode (R0's limbs and the storage states are identically zero there)
W_COMMITTED_DEFERRED = 168  # deferred mode: only R0's limbs and state are left out
W_AUX = 40                  # aux trace of the lookup argument: H0..H7, HR, S as four base columns each
RC_TABLE = 1024
N_LK = 56                   # lookup parameters: alpha (4), lambda^0..11 (48), T / N (4)
HEADER_WORDS = 157
_bound = False


def lib():
    global _bound
    L = api.lib()
    if not _bound:
        V, U32, SZ, I = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
        L.so_p.restype = U32
        L.so_fmul.restype = U32; L.so_fmul.argtypes = [U32, U32]
        L.so_finv.restype = U32; L.so_finv.argtypes = [U32]
        L.so_root_of_unity.restype = U32; L.so_root_of_unity.argtypes = [I]
        PP = C.POINTER(PublicC)
        for name, args in [("so_emul", [V, V, V]), ("so_einv", [V, V]), ("so_poseidon2_permute", [V]), ("so_poseidon2_constants", [V, V, V]),
                           ("so_hash_elems", [V, SZ, V]), ("so_compress", [V, V, V]), ("so_digest_bytes", [V, SZ, V]), ("so_ntt", [V, SZ, I]),
                           ("so_lde", [V, SZ, I, V, V]), ("so_main_trace", [V, PP, V]), ("so_merkle", [V, I, SZ, V, V]),
                           ("so_commit_trace", [V, PP, I, V, V])]:
            f = getattr(L, name); f.restype = None; f.argtypes = args
        L.so_commit_trace_blocked.restype = None; L.so_commit_trace_blocked.argtypes = [V, PP, I, V, I]
        L.so_main_trace_width.restype = I
        L.so_committed_width.restype = I; L.so_committed_width.argtypes = [I]
        L.so_to_committed.restype = None; L.so_to_committed.argtypes = [V, SZ, I, V]
        L.so_padded_log_n.restype = I; L.so_padded_log_n.argtypes = [C.c_uint64]
        L.so_num_constraints.restype = I
        L.so_constraints_eval.restype = I; L.so_constraints_eval.argtypes = [V, V, V, V, V, U32, U32, U32, PP, V, V]
        L.so_lookup_setup.restype = SZ; L.so_lookup_setup.argtypes = [V, PP, V, V, V, V, V, V]
        L.so_aux_width.restype = I; L.so_rc_table.restype = I
        L.so_last_lookup.restype = None; L.so_last_lookup.argtypes = [V]
        L.so_last_aux.restype = None; L.so_last_aux.argtypes = [V]
        L.so_prove.restype = SZ; L.so_prove.argtypes = [V, PP, V, SZ]
        L.so_prove_matrix.restype = SZ; L.so_prove_matrix.argtypes = [V, PP, V, SZ]
        L.so_prove_lean.restype = SZ; L.so_prove_lean.argtypes = [V, PP, V, SZ, I]
        L.so_verify.restype = I; L.so_verify.argtypes = [V, SZ, PP]
        L.so_pow_bits.restype = I; L.so_header_words.restype = I; L.so_state_words.restype = I
        L.so_verify_segment.restype = I; L.so_verify_segment.argtypes = [V, SZ, PP, V]
        L.so_verify_chain.restype = I; L.so_verify_chain.argtypes = [V, V, I, PP]
        L.so_constraints_eval_states.restype = I; L.so_constraints_eval_states.argtypes = [V, V, V, V, V, U32, U32, U32, PP, V, V, V, V]
        L.so_last_challenges.restype = None; L.so_last_challenges.argtypes = [V, V, V]
        L.so_last_quotient.restype = None; L.so_last_quotient.argtypes = [V]
        L.so_last_fri_layer.restype = SZ; L.so_last_fri_layer.argtypes = [I, V]
        L.so_num_queries.restype = I; L.so_log_final.restype = I
        _bound = True
    return L


class PublicC(C.Structure):
    """so_public: the public inputs of a proof (observed first by the transcript, carried in the proof header)."""
    _fields_ = [("n_real", C.c_uint64), ("deferred", C.c_uint32), ("fri", C.c_uint32), ("entry", C.c_uint64), ("prog", C.c_uint32 * 4), ("io", C.c_uint32 * 4),
                ("blob", C.c_char_p), ("blob_len", C.c_uint64),      # the program itself (prover side): its code words are the instruction ROM
                # mode 2 (`deferred` == 2: the default VM mode WITH the I/O argument): the tapes and the halt reason in the clear, and for a SEGMENT the
                # WRITE / READ ecalls the run executed before its first row
                ("inputs", C.c_void_p), ("n_inputs", C.c_uint64), ("outputs", C.c_void_p), ("n_outputs", C.c_uint64), ("halt_kind", C.c_uint32), ("pad2", C.c_uint32),
                ("halt_code", C.c_uint64), ("writes_before", C.c_uint64), ("reads_before", C.c_uint64)]

    def set_io(self, inputs, outputs, halt, writes_before=0, reads_before=0):
        self._in_ref = np.ascontiguousarray(np.asarray([int(x) & (2**64 - 1) for x in inputs], dtype=np.uint64))
        self._out_ref = np.ascontiguousarray(np.asarray([int(x) & (2**64 - 1) for x in outputs], dtype=np.uint64))
        self.inputs, self.n_inputs = (self._in_ref.ctypes.data if len(self._in_ref) else None), len(self._in_ref)
        self.outputs, self.n_outputs = (self._out_ref.ctypes.data if len(self._out_ref) else None), len(self._out_ref)
        self.halt_kind, self.halt_code = int(halt[0]), int(halt[1]) if int(halt[0]) == 1 else 0
        self.writes_before, self.reads_before = int(writes_before), int(reads_before)
        return self

    def set_blob(self, blob: bytes):
        self._blob_ref = bytes(blob)            # keeps the bytes alive as long as the struct
        self.blob = self._blob_ref
        self.blob_len = len(self._blob_ref)
        return self

    def clone(self):
        q = PublicC(self.n_real, self.deferred, self.fri, self.entry)
        q.prog[:] = list(self.prog); q.io[:] = list(self.io)
        if hasattr(self, "_in_ref"):
            q.set_io(self._in_ref.tolist(), self._out_ref.tolist(), (self.halt_kind, self.halt_code), self.writes_before, self.reads_before)
        return q.set_blob(getattr(self, "_blob_ref", b""))


def digest_bytes(b: bytes) -> np.ndarray:
    o = np.zeros(4, np.uint32)
    lib().so_digest_bytes(bytes(b), len(b), o.ctypes.data)
    return o


def io_bytes(inputs, outputs, halt_kind: int, halt_code: int, cycles: int) -> bytes:
    """What the io digest covers: little-endian u64 words [n_inputs, inputs.., n_outputs, outputs.., halt kind, halt code, cycles]."""
    words = [len(inputs), *inputs, len(outputs), *outputs, halt_kind, halt_code, cycles]
    return np.array([int(x) & (2**64 - 1) for x in words], dtype="<u8").tobytes()


def public_inputs(n_real: int, blob: bytes = b"", inputs=(), outputs=(), halt=(2, 0), deferred: bool = False, entry: int | None = None, io_mode: bool = False,
                  writes_before: int = 0, reads_before: int = 0, mem_mode: bool = False, num_queries: int = 0, pow_bits: int = 0, wide_mode: bool = False) -> PublicC:
    """Public inputs of a run: halt = (kind, code) with kind 0 Ebreak / 1 Exit / 2 CycleLimit; entry defaults to the blob header's.
    io_mode = mode 2: the default VM mode with the I/O argument (the proof carries the tapes; WRITE / READ ecalls are tied to them).
    mem_mode = mode 3: mode 2 with the memory argument (loads and stores constrained, every access tied to a consistent memory; the proof carries the touched cells).
    wide_mode = mode 4 (round 6): mode 3 with MULH / DIVU / REMU / DIV / REM on operands below 2^40 (six more range slots per row)."""
    if entry is None:
        entry = int.from_bytes(blob[12:16], "little") if len(blob) >= 16 else 0x1000
    assert not (deferred and (io_mode or mem_mode or wide_mode)), "the I/O and memory arguments are stated for the default VM mode"
    p = PublicC(n_real, 4 if wide_mode else 3 if mem_mode else 2 if io_mode else int(deferred), (int(num_queries) & 0xFFFF) | (int(pow_bits) << 16), entry)      # fri: the prover's parameters, 0 = 50 queries + 12 bits
    p.set_blob(blob)
    p.set_io(list(inputs), list(outputs), halt, writes_before, reads_before)
    p.prog[:] = [int(x) for x in digest_bytes(blob)]
    p.io[:] = [int(x) for x in digest_bytes(io_bytes(list(inputs), list(outputs), halt[0], halt[1], n_real))]
    return p


def padded_log_n(n_real: int) -> int:
    return lib().so_padded_log_n(n_real)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def emul(a, b):
    a, b, o = _u32(a), _u32(b), np.zeros(4, np.uint32)
    lib().so_emul(a.ctypes.data, b.ctypes.data, o.ctypes.data)
    return o


def einv(a):
    a, o = _u32(a), np.zeros(4, np.uint32)
    lib().so_einv(a.ctypes.data, o.ctypes.data)
    return o


def permute(state):
    s = _u32(state).copy()
    lib().so_poseidon2_permute(s.ctypes.data)
    return s


def constants():
    ext, inn, diag = np.zeros((8, 12), np.uint32), np.zeros(22, np.uint32), np.zeros(12, np.uint32)
    lib().so_poseidon2_constants(ext.ctypes.data, inn.ctypes.data, diag.ctypes.data)
    return ext, inn, diag


def hash_elems(x):
    x, o = _u32(x), np.zeros(4, np.uint32)
    lib().so_hash_elems(x.ctypes.data, len(x), o.ctypes.data)
    return o


def compress(l, r):
    l, r, o = _u32(l), _u32(r), np.zeros(4, np.uint32)
    lib().so_compress(l.ctypes.data, r.ctypes.data, o.ctypes.data)
    return o


def ntt(a, inverse=False):
    a = _u32(a).copy()
    lib().so_ntt(a.ctypes.data, len(a), int(inverse))
    return a


def lde(evals, log_blowup=1):
    e = _u32(evals)
    coeffs, out = np.zeros(len(e), np.uint32), np.zeros(len(e) << log_blowup, np.uint32)
    lib().so_lde(e.ctypes.data, len(e), log_blowup, coeffs.ctypes.data, out.ctypes.data)
    return coeffs, out


def _pub(rows, pub):
    return pub if pub is not None else public_inputs(len(rows))


def main_trace(rows: np.ndarray, pub: PublicC | None = None) -> np.ndarray:
    """Packed reference rows (api.ROW_DTYPE) -> Baby Bear matrix [W_MAIN][N], N = the padded power of two."""
    rows = np.ascontiguousarray(rows)
    pub = _pub(rows, pub)
    out = np.zeros((logical_width(pub.deferred), 1 << padded_log_n(pub.n_real)), np.uint32)
    lib().so_main_trace(rows.ctypes.data, C.byref(pub), out.ctypes.data)
    return out


def logical_width(mode=0) -> int:
    """Logical main-trace columns of a mode (0 default, 1 deferred, 2 default + I/O, 3 default + I/O + memory): 172 / 172 / 180 / 220."""
    return lib().so_logical_width(int(mode))


def aux_width(mode=0) -> int:
    return lib().so_aux_width_for(int(mode))


def committed_width(deferred=False) -> int:
    """Committed columns of a mode (a bool reads as mode 0 / 1; 2 = default + I/O): 152 / 168 / 160."""
    return lib().so_committed_width(int(deferred))


def to_committed(matrix: np.ndarray, deferred=False) -> np.ndarray:
    """Logical main-trace matrix [logical_width][n] -> the committed one [committed_width][n] (so::to_physical); `deferred` = the mode."""
    m = _u32(matrix)
    out = np.zeros((committed_width(deferred), m.shape[1]), np.uint32)
    lib().so_to_committed(m.ctypes.data, m.shape[1], int(deferred), out.ctypes.data)
    return out


def lookup_setup(matrix: np.ndarray, pub: PublicC, alpha_l, lam):
    """The lookup side of a main-trace matrix for GIVEN challenges: (aux trace [W_AUX][N], lk[56] = alpha, lambda powers, T / N,
    ROM multiplicities, range multiplicities)."""
    m, a, l = _u32(matrix), _u32(alpha_l), _u32(lam)
    n = m.shape[1]
    aux, lk, rc = np.zeros((W_AUX, n), np.uint32), np.zeros(N_LK, np.uint32), np.zeros(RC_TABLE, np.uint32)
    n_rom = lib().so_lookup_setup(m.ctypes.data, C.byref(pub), a.ctypes.data, l.ctypes.data, None, None, None, None)
    rom = np.zeros(max(n_rom, 1), np.uint32)
    lib().so_lookup_setup(m.ctypes.data, C.byref(pub), a.ctypes.data, l.ctypes.data, aux.ctypes.data, lk.ctypes.data, rom.ctypes.data, rc.ctypes.data)
    return aux, lk, rom[:n_rom], rc


def constraints_eval(loc, nxt, aloc, anxt, lk, is_first, is_last, is_trans, pub: PublicC, alpha) -> np.ndarray:
    """Σ alpha^c C_c of one (local, next) row pair — main columns, aux columns, lookup parameters — for the given selector values (E4 result)."""
    loc, nxt, aloc, anxt, lk, alpha, o = _u32(loc), _u32(nxt), _u32(aloc), _u32(anxt), _u32(lk), _u32(alpha), np.zeros(4, np.uint32)
    lib().so_constraints_eval(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(is_first), int(is_last), int(is_trans), C.byref(pub),
                              alpha.ctypes.data, o.ctypes.data)
    return o


def merkle(mat: np.ndarray, want_layers=False):
    mat = _u32(mat)
    w, n = mat.shape
    root = np.zeros(4, np.uint32)
    layers = np.zeros(4 * (2 * n - 1), np.uint32) if want_layers else None
    lib().so_merkle(mat.ctypes.data, w, n, root.ctypes.data, layers.ctypes.data if want_layers else None)
    return (root, layers) if want_layers else root


def commit_trace(rows: np.ndarray, log_blowup=1, want_lde=False, pub: PublicC | None = None):
    rows = np.ascontiguousarray(rows)
    pub = _pub(rows, pub)
    n = 1 << padded_log_n(pub.n_real)
    root = np.zeros(4, np.uint32)
    L = np.zeros((committed_width(pub.deferred), n << log_blowup), np.uint32) if want_lde else None
    lib().so_commit_trace(rows.ctypes.data, C.byref(pub), log_blowup, root.ctypes.data, L.ctypes.data if want_lde else None)
    return (root, L) if want_lde else root


def commit_trace_blocked(rows: np.ndarray, log_blowup=1, pub: PublicC | None = None, threads=1):
    """The root of commit_trace, computed eight columns at a time with one sponge state per leaf (so_commit_trace_blocked): what makes 2^24 rows fit."""
    rows = np.ascontiguousarray(rows)
    pub = _pub(rows, pub)
    root = np.zeros(4, np.uint32)
    lib().so_commit_trace_blocked(rows.ctypes.data, C.byref(pub), log_blowup, root.ctypes.data, int(threads))
    return root


# ---- stage B: prover + verifier ------------------------------------------------------------------
def prove(rows: np.ndarray, pub: PublicC | None = None) -> np.ndarray:
    """Full ZKIR-STARK v1 proof (u32 words) of the executed rows (any count >= 1; padded to a power of two >= 8)."""
    rows = np.ascontiguousarray(rows)
    pub = _pub(rows, pub)
    assert pub.n_real == len(rows) >= 1
    size = lib().so_prove(rows.ctypes.data, C.byref(pub), None, 0)
    out = np.zeros(size, np.uint32)
    lib().so_prove(rows.ctypes.data, C.byref(pub), out.ctypes.data, size)
    return out


def prove_lean(rows: np.ndarray, pub: PublicC | None = None, threads: int = 1, cap_words: int = 1 << 22) -> np.ndarray:
    """The proof of `prove`, by the memory-lean threaded restatement (so::prove_lean: no committed copy, no coefficient vectors, std::threads over columns / leaves / coset
    points): what makes the whole proof at 2^24 rows fit this container (38 GB against ~70).  One pass: the buffer must hold the proof (a 2^24-row mode-0 proof is ~0.1 M words)."""
    rows = np.ascontiguousarray(rows)
    pub = _pub(rows, pub)
    assert pub.n_real == len(rows) >= 1
    out = np.zeros(cap_words, np.uint32)
    size = lib().so_prove_lean(rows.ctypes.data, C.byref(pub), out.ctypes.data, cap_words, int(threads))
    assert size <= cap_words, f"proof of {size} words does not fit cap_words = {cap_words}"
    return out[:size].copy()


def prove_matrix(matrix: np.ndarray, pub: PublicC) -> np.ndarray:
    """Proof of a GIVEN main-trace matrix [W_MAIN][N] (tests: what a cheating prover would submit)."""
    m = _u32(matrix)
    assert m.shape == (logical_width(pub.deferred), 1 << padded_log_n(pub.n_real))
    size = lib().so_prove_matrix(m.ctypes.data, C.byref(pub), None, 0)
    out = np.zeros(size, np.uint32)
    lib().so_prove_matrix(m.ctypes.data, C.byref(pub), out.ctypes.data, size)
    return out


def mem_cells(rows: np.ndarray, pub: PublicC) -> np.ndarray:
    """(mode 3) the touched memory cells of a run, [n][7] words as the proof carries them: address limbs (20 + 20 bits), time of the last access, final bytes (4 x 16 bits)."""
    rows = np.ascontiguousarray(rows)
    L = lib()
    L.so_mem_cells.restype = C.c_size_t; L.so_mem_cells.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    n = L.so_mem_cells(rows.ctypes.data, C.byref(pub), None, 0)
    out = np.zeros((max(n, 1), 7), np.uint32)
    L.so_mem_cells(rows.ctypes.data, C.byref(pub), out.ctypes.data, n)
    return out[:n]


def hash_section(rows: np.ndarray, pub: PublicC) -> np.ndarray:
    """(mode 4) the hash calls of a run as the proof carries them (so::hash_section): [n] then per call 8 words + 5 per touched cell."""
    rows = np.ascontiguousarray(rows)
    L = lib()
    L.so_hash_section.restype = C.c_size_t; L.so_hash_section.argtypes = [C.c_void_p, C.POINTER(PublicC), C.c_void_p, C.c_size_t]
    n = L.so_hash_section(rows.ctypes.data, C.byref(pub), None, 0)
    out = np.zeros(n, np.uint32)
    L.so_hash_section(rows.ctypes.data, C.byref(pub), out.ctypes.data, n)
    return out


def set_hash_calls(words=None) -> None:
    """(tests, mode 4) the hash section prove_matrix_mem / failing_constraints are to use (a matrix brings no rows to replay); None clears."""
    L = lib()
    L.so_set_hash_calls.restype = None; L.so_set_hash_calls.argtypes = [C.c_void_p, C.c_size_t]
    if words is None or len(words) == 0:
        L.so_set_hash_calls(None, 0)
    else:
        w = _u32(words)
        L.so_set_hash_calls(w.ctypes.data, len(w))


def prove_matrix_mem(matrix: np.ndarray, pub: PublicC, cells: np.ndarray) -> np.ndarray:
    """(mode 3) proof of a GIVEN main-trace matrix with a GIVEN list of touched cells (tests: a cheating prover)."""
    m, c = _u32(matrix), _u32(cells).reshape(-1, 7)
    assert m.shape == (logical_width(pub.deferred), 1 << padded_log_n(pub.n_real)) and pub.deferred >= 3
    L = lib()
    L.so_prove_matrix_mem.restype = C.c_size_t; L.so_prove_matrix_mem.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    size = L.so_prove_matrix_mem(m.ctypes.data, C.byref(pub), c.ctypes.data, len(c), None, 0)
    out = np.zeros(size, np.uint32)
    L.so_prove_matrix_mem(m.ctypes.data, C.byref(pub), c.ctypes.data, len(c), out.ctypes.data, size)
    return out


def failing_constraints(matrix: np.ndarray, pub: PublicC, cells=None, alpha_l=(3, 1, 4, 1), lam=(2, 7, 1, 8), cap: int = 64):
    """(tests) the (constraint index, row) pairs a main-trace matrix violates on the trace domain, with the lookup side set up honestly for the given challenges
    (any mode; mode 3 takes the touched cells).  The running-sum constraints (the last four of the base list) fail on the wrap-around row exactly when the LogUp sums differ."""
    m = _u32(matrix)
    c = _u32(cells if cells is not None else np.zeros((0, 7))).reshape(-1, 7)
    a, l, out = _u32(alpha_l), _u32(lam), np.zeros((cap, 2), np.uint32)
    L = lib()
    L.so_failing_constraints.restype = C.c_size_t
    L.so_failing_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    n = L.so_failing_constraints(m.ctypes.data, C.byref(pub), c.ctypes.data if len(c) else None, len(c), a.ctypes.data, l.ctypes.data, out.ctypes.data, cap)
    return n, [tuple(int(x) for x in r) for r in out[:min(n, cap)]]


def verify(proof: np.ndarray, expect: PublicC | None = None) -> int:
    """0 = accepted; otherwise the code of the first failed check (6: the header's public inputs are not the expected ones)."""
    proof = _u32(proof)
    return lib().so_verify(proof.ctypes.data, len(proof), C.byref(expect) if expect is not None else None)


def verify_segment(proof: np.ndarray, expect: PublicC | None = None):
    """A SEGMENT of a run (no initial-state requirement): (code, first_state[68], last_state[68]) — what verify_chain links."""
    proof = _u32(proof)
    st = np.zeros(136, np.uint32)
    rc = lib().so_verify_segment(proof.ctypes.data, len(proof), C.byref(expect) if expect is not None else None, st.ctypes.data)
    return rc, st[:68].copy(), st[68:].copy()


def _u64(a):
    return np.ascontiguousarray(np.asarray(list(a), dtype=np.uint64))


def verify_io(proof: np.ndarray, expect: PublicC | None, inputs, outputs, halt) -> int:
    """verify + the claim in the clear: (inputs, outputs, halt = (kind, code)) hash to the proof's io digest with its row count (50) and the halt row is the
    instruction the halt reason names (52) with the claimed exit code (53)."""
    proof, i, o = _u32(proof), _u64(inputs), _u64(outputs)
    L = lib()
    L.so_verify_io.restype = C.c_int
    L.so_verify_io.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint64]
    return L.so_verify_io(proof.ctypes.data, len(proof), C.byref(expect) if expect is not None else None, i.ctypes.data, len(i), o.ctypes.data, len(o), int(halt[0]), int(halt[1]))


def verify_chain_io(proofs, expect: PublicC | None, inputs, outputs, halt) -> int:
    ps = [_u32(p) for p in proofs]
    ptrs = (C.c_void_p * len(ps))(*[p.ctypes.data for p in ps])
    lens = (C.c_size_t * len(ps))(*[len(p) for p in ps])
    i, o = _u64(inputs), _u64(outputs)
    L = lib()
    L.so_verify_chain_io.restype = C.c_int
    L.so_verify_chain_io.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint64]
    return L.so_verify_chain_io(ptrs, lens, len(ps), C.byref(expect) if expect is not None else None, i.ctypes.data, len(i), o.ctypes.data, len(o), int(halt[0]), int(halt[1]))


def verify_chain(proofs, expect: PublicC | None = None) -> int:
    """Segments of one run in order (each overlapping its predecessor by one row).  0 = accepted; 40-44 chain checks; 1000 (i+1) + c = check c
    of segment i.  `expect.n_real` = the run's total executed rows."""
    ps = [_u32(p) for p in proofs]
    ptrs = (C.c_void_p * len(ps))(*[p.ctypes.data for p in ps])
    lens = (C.c_size_t * len(ps))(*[len(p) for p in ps])
    return lib().so_verify_chain(ptrs, lens, len(ps), C.byref(expect) if expect is not None else None)


def constraints_eval_states(loc, nxt, aloc, anxt, lk, is_first, is_last, is_trans, pub: PublicC, first, last, alpha) -> np.ndarray:
    loc, nxt, aloc, anxt, lk, alpha, first, last, o = _u32(loc), _u32(nxt), _u32(aloc), _u32(anxt), _u32(lk), _u32(alpha), _u32(first), _u32(last), np.zeros(4, np.uint32)
    lib().so_constraints_eval_states(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(is_first), int(is_last), int(is_trans),
                                     C.byref(pub), first.ctypes.data, last.ctypes.data, alpha.ctypes.data, o.ctypes.data)
    return o


def proof_layout(proof: np.ndarray) -> dict:
    """Word offsets of the parts of a proof (format v5 on) that follow the header: program, multiplicities, roots, openings."""
    wt = int(proof[3]) + W_AUX                                   # header word 3 = committed main-trace columns
    blob_len = int(proof[HEADER_WORDS])
    at = HEADER_WORDS + 1
    blob_words = (blob_len + 1) // 2
    blob = bytearray(blob_len)
    for i in range(0, blob_len, 2):
        h = int(proof[at + i // 2])
        blob[i] = h & 0xFF
        if i + 1 < blob_len:
            blob[i + 1] = h >> 8
    n_rom = int.from_bytes(blob[16:20], "little") // 4 if blob_len >= 32 else 0
    rom_mult = at + blob_words
    rc_mult = rom_mult + n_rom
    troot = rc_mult + RC_TABLE
    return {"blob_len": blob_len, "blob": bytes(blob), "blob_at": at, "n_rom": n_rom, "rom_mult": rom_mult, "rc_mult": rc_mult, "trace_root": troot, "aux_root": troot + 4,
            "quotient_root": troot + 8, "openings": troot + 12, "t_z": troot + 12, "t_zw": troot + 12 + 4 * wt, "q_z": troot + 12 + 8 * wt}


STATE_COLS = [0, 1, 2, 3] + list(range(9, 73))      # cycle, pc limbs, register limbs, storage states (so::state_col)


def last_challenges():
    a, z, g = np.zeros(4, np.uint32), np.zeros(4, np.uint32), np.zeros(4, np.uint32)
    lib().so_last_challenges(a.ctypes.data, z.ctypes.data, g.ctypes.data)
    return a, z, g


def last_quotient(n: int) -> np.ndarray:
    out = np.zeros((4, 2 * n), np.uint32)
    lib().so_last_quotient(out.ctypes.data)
    return out


def last_fri_layer(j: int) -> np.ndarray:
    m = lib().so_last_fri_layer(j, None)
    out = np.zeros((m, 4), np.uint32)
    lib().so_last_fri_layer(j, out.ctypes.data)
    return out
