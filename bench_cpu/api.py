"""ctypes loader of bench_cpu/libzkir_cpu_port.so — the multi-threaded CPU port of the commit stage (LDE + Poseidon2 Merkle).

Used by bench.py's `cpu_baseline.commit_stage_self_defined` side figure and by one CPU test that ties its root to the naive oracle's.
It contains the product's field / hash headers compiled for the host, which is why it lives outside oracle/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzkir_cpu_port.so")
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = C.CDLL(_SO)
        L.so_commit_port.restype = None
        L.so_commit_port.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def commit_port(matrix: np.ndarray, threads: int):
    """Column-major main-trace matrix [W][N] (canonical u32) -> (root u32[4], lde_seconds, merkle_seconds) on `threads` host threads."""
    m = np.ascontiguousarray(matrix, dtype=np.uint32)
    w, n = m.shape
    root, secs = np.zeros(4, np.uint32), np.zeros(2, np.float64)
    lib().so_commit_port(m.ctypes.data, w, int(n).bit_length() - 1, int(threads), root.ctypes.data, secs.ctypes.data)
    return root, float(secs[0]), float(secs[1])
