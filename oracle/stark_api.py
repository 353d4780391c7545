"""ctypes view of oracle/stark_oracle.cpp (self-defined prover stages; PARITY UNPINNED).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api

P = 2013265921
W_MAIN = 89
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
        for name, args in [("so_emul", [V, V, V]), ("so_einv", [V, V]), ("so_poseidon2_permute", [V]), ("so_poseidon2_constants", [V, V, V]),
                           ("so_hash_elems", [V, SZ, V]), ("so_compress", [V, V, V]), ("so_ntt", [V, SZ, I]), ("so_lde", [V, SZ, I, V, V]),
                           ("so_main_trace", [V, SZ, V]), ("so_merkle", [V, I, SZ, V, V]), ("so_commit_trace", [V, SZ, I, V, V])]:
            f = getattr(L, name); f.restype = None; f.argtypes = args
        L.so_main_trace_width.restype = I
        L.so_prove.restype = SZ; L.so_prove.argtypes = [V, SZ, V, SZ]
        L.so_verify.restype = I; L.so_verify.argtypes = [V, SZ]
        L.so_last_challenges.restype = None; L.so_last_challenges.argtypes = [V, V, V]
        L.so_last_quotient.restype = None; L.so_last_quotient.argtypes = [V]
        L.so_last_fri_layer.restype = SZ; L.so_last_fri_layer.argtypes = [I, V]
        L.so_num_queries.restype = I; L.so_log_final.restype = I
        _bound = True
    return L


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


def main_trace(rows: np.ndarray) -> np.ndarray:
    """Packed reference rows (api.ROW_DTYPE) -> Baby Bear matrix [W_MAIN][n]."""
    rows = np.ascontiguousarray(rows)
    out = np.zeros((W_MAIN, len(rows)), np.uint32)
    lib().so_main_trace(rows.ctypes.data, len(rows), out.ctypes.data)
    return out


def merkle(mat: np.ndarray, want_layers=False):
    mat = _u32(mat)
    w, n = mat.shape
    root = np.zeros(4, np.uint32)
    layers = np.zeros(4 * (2 * n - 1), np.uint32) if want_layers else None
    lib().so_merkle(mat.ctypes.data, w, n, root.ctypes.data, layers.ctypes.data if want_layers else None)
    return (root, layers) if want_layers else root


def commit_trace(rows: np.ndarray, log_blowup=1, want_lde=False):
    rows = np.ascontiguousarray(rows)
    n = len(rows)
    root = np.zeros(4, np.uint32)
    L = np.zeros((W_MAIN, n << log_blowup), np.uint32) if want_lde else None
    lib().so_commit_trace(rows.ctypes.data, n, log_blowup, root.ctypes.data, L.ctypes.data if want_lde else None)
    return (root, L) if want_lde else root


# ---- stage B: prover + verifier ------------------------------------------------------------------
def prove(rows: np.ndarray) -> np.ndarray:
    """Full ZKIR-STARK v0 proof (u32 words) for a power-of-two number of packed reference rows."""
    rows = np.ascontiguousarray(rows)
    n = len(rows)
    assert n & (n - 1) == 0 and n >= 8
    size = lib().so_prove(rows.ctypes.data, n, None, 0)
    out = np.zeros(size, np.uint32)
    lib().so_prove(rows.ctypes.data, n, out.ctypes.data, size)
    return out


def verify(proof: np.ndarray) -> int:
    """0 = accepted; otherwise the code of the first failed check."""
    proof = _u32(proof)
    return lib().so_verify(proof.ctypes.data, len(proof))


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
