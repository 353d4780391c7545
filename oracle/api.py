"""ctypes view of the CPU oracle (oracle/zkir_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under zkir_amd/ does.  It never touches the GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libzkir_oracle.so")

# packed record layouts (must match the #pragma pack(1) structs in zkir_oracle.cpp)
ROW_DTYPE = np.dtype([
    ("cycle", "<u8"), ("pc", "<u8"), ("instruction", "<u4"),
    ("registers", "<u8", (16,)), ("bound_bits", "<u4", (16,)), ("bound_tag", "u1", (16,)),
    ("bound_payload", "<u8", (16,)), ("reg_state", "u1", (16,)),
])
MEMOP_DTYPE = np.dtype([
    ("address", "<u8"), ("value", "<u8"), ("timestamp", "<u8"), ("is_write", "u1"), ("width", "u1"),
    ("bound_bits", "<u4"), ("bound_tag", "u1"), ("bound_payload", "<u8"),
])
RC_DTYPE = np.dtype([("value", "<u8"), ("pc", "<u8"), ("chunks", "<u2", (4,))])
NORM_DTYPE = np.dtype([
    ("cycle", "<u8"), ("pc", "<u8"), ("reg", "u1"), ("accumulated", "<u8", (2,)), ("normalized", "<u4", (2,)),
    ("carries", "<u4", (2,)), ("normalized_bits", "u1"), ("limb_bits", "u1"), ("cause", "u1"), ("opcode", "u1"),
])
assert ROW_DTYPE.itemsize == 372 and MEMOP_DTYPE.itemsize == 39 and RC_DTYPE.itemsize == 24 and NORM_DTYPE.itemsize == 53

HALT_EBREAK, HALT_EXIT, HALT_CYCLE_LIMIT = 0, 1, 2


class _Cfg(C.Structure):
    _fields_ = [("max_cycles", C.c_uint64), ("trace", C.c_uint8), ("enable_range_checking", C.c_uint8),
                ("enable_execution_trace", C.c_uint8), ("enable_deferred_model", C.c_uint8)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("zkir_oracle.cpp", "stark_oracle.cpp", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.zo_run.restype = C.c_void_p
        L.zo_run.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(_Cfg), C.c_int]
        L.zo_run_window.restype = C.c_void_p
        L.zo_run_window.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(_Cfg), C.c_uint64, C.c_uint64]
        for name, res in [("zo_error_code", C.c_int), ("zo_error_msg", C.c_char_p), ("zo_cycles", C.c_uint64),
                          ("zo_halt_kind", C.c_int), ("zo_halt_code", C.c_uint64), ("zo_n_outputs", C.c_size_t),
                          ("zo_outputs", C.c_void_p), ("zo_n_rows", C.c_size_t), ("zo_rows", C.c_void_p),
                          ("zo_n_memops", C.c_size_t), ("zo_memops", C.c_void_p), ("zo_row_memop_offsets", C.c_void_p),
                          ("zo_n_rc_witnesses", C.c_size_t), ("zo_rc_offsets", C.c_void_p), ("zo_n_rc_checks", C.c_size_t),
                          ("zo_rc_checks", C.c_void_p), ("zo_n_norm_events", C.c_size_t), ("zo_norm_events", C.c_void_p)]:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = [C.c_void_p]
        L.zo_free.argtypes = [C.c_void_p]
        L.zo_free.restype = None
        L.zo_sorted_memops.argtypes = [C.c_void_p, C.c_void_p]
        L.zo_sorted_memops.restype = None
        L.zo_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.zo_keccak256.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.zo_blake3.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.zo_sha256_witness.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_void_p]
        L.zo_sha256_witness.restype = C.c_int
        for name in ["zo_m31_add", "zo_m31_sub", "zo_m31_mul", "zo_m31_pow"]:
            getattr(L, name).argtypes = [C.c_uint32, C.c_uint32]
            getattr(L, name).restype = C.c_uint32
        for name in ["zo_m31_neg", "zo_m31_inv", "zo_m31_new"]:
            getattr(L, name).argtypes = [C.c_uint32]
            getattr(L, name).restype = C.c_uint32
        L.zo_decode.argtypes = [C.c_uint32] + [C.c_void_p] * 6
        L.zo_decode.restype = C.c_uint32
        L.zo_value40_op.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _copy(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class OracleError(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code, self.msg = code, msg


@dataclass
class OracleResult:
    cycles: int
    halt_kind: int
    halt_code: int
    outputs: np.ndarray
    rows: np.ndarray              # ROW_DTYPE[n_rows]
    memops: np.ndarray            # MEMOP_DTYPE[n_memops], row order
    row_memop_offsets: np.ndarray  # u64[n_rows+1]
    sorted_memops: np.ndarray     # ExecutionResult::get_memory_trace()
    rc_offsets: np.ndarray        # u64[n_witnesses+1]
    rc_checks: np.ndarray         # RC_DTYPE
    norm_events: np.ndarray       # NORM_DTYPE


def run(program_blob: bytes, inputs=(), max_cycles: int = 1_000_000, enable_range_checking: bool = False,
        enable_execution_trace: bool = False, enable_deferred_model: bool = False, faithful: bool = False,
        want_sorted: bool = True, keep_rows: tuple | None = None) -> OracleResult:
    """VM::new(program, inputs, config).run() of the reference (vm.rs:138-358), restated on the CPU.
    keep_rows=(lo, hi): run everything but keep only the rows / memory ops of cycles [lo, hi) (tests at 2^22..2^26 rows)."""
    L = lib()
    cfg = _Cfg(max_cycles, 0, int(enable_range_checking), int(enable_execution_trace), int(enable_deferred_model))
    arr = (C.c_uint64 * max(1, len(inputs)))(*inputs)
    if keep_rows is not None:
        h = L.zo_run_window(program_blob, len(program_blob), arr, len(inputs), C.byref(cfg), int(keep_rows[0]), int(keep_rows[1]))
    else:
        h = L.zo_run(program_blob, len(program_blob), arr, len(inputs), C.byref(cfg), int(faithful))
    try:
        code = L.zo_error_code(h)
        if code != 0:
            raise OracleError(code, L.zo_error_msg(h).decode())
        n_rows = L.zo_n_rows(h)
        n_mem = L.zo_n_memops(h)
        n_w = L.zo_n_rc_witnesses(h)
        sorted_ops = np.zeros(n_mem, dtype=MEMOP_DTYPE)
        if want_sorted and n_mem:
            L.zo_sorted_memops(h, sorted_ops.ctypes.data)
        return OracleResult(
            cycles=L.zo_cycles(h), halt_kind=L.zo_halt_kind(h), halt_code=L.zo_halt_code(h),
            outputs=_copy(L.zo_outputs(h), L.zo_n_outputs(h), "<u8"),
            rows=_copy(L.zo_rows(h), n_rows, ROW_DTYPE),
            memops=_copy(L.zo_memops(h), n_mem, MEMOP_DTYPE),
            row_memop_offsets=_copy(L.zo_row_memop_offsets(h), n_rows + 1, "<u8") if n_rows or True else None,
            sorted_memops=sorted_ops,
            rc_offsets=_copy(L.zo_rc_offsets(h), n_w + 1, "<u8"),
            rc_checks=_copy(L.zo_rc_checks(h), L.zo_n_rc_checks(h), RC_DTYPE),
            norm_events=_copy(L.zo_norm_events(h), L.zo_n_norm_events(h), NORM_DTYPE),
        )
    finally:
        L.zo_free(h)


def time_run(program_blob: bytes, max_cycles: int, faithful: bool = False, build_only: bool = False) -> tuple[float, int]:
    """Wall time of one trace-generating run (no result copies): (seconds, rows).  build_only: every row is built (pre-state copy,
    memory-op filter) but not kept — for sizes whose 372 B/row would not fit the host's memory; returns the cycle count as rows."""
    import time
    L = lib()
    cfg = _Cfg(max_cycles, 0, 0, 1, 0)
    arr = (C.c_uint64 * 1)()
    t0 = time.perf_counter()
    if build_only:
        h = L.zo_run_window(program_blob, len(program_blob), arr, 0, C.byref(cfg), max_cycles + 1, max_cycles + 2)
    else:
        h = L.zo_run(program_blob, len(program_blob), arr, 0, C.byref(cfg), int(faithful))
    dt = time.perf_counter() - t0
    n = L.zo_cycles(h) if build_only else L.zo_n_rows(h)
    L.zo_free(h)
    return dt, n


def sha256(data: bytes) -> np.ndarray:
    out = np.zeros(8, dtype="<u4")
    lib().zo_sha256(data, len(data), out.ctypes.data)
    return out


def keccak256(data: bytes) -> bytes:
    out = np.zeros(32, dtype="u1")
    lib().zo_keccak256(data, len(data), out.ctypes.data)
    return out.tobytes()


def blake3(data: bytes) -> bytes:
    out = np.zeros(32, dtype="u1")
    lib().zo_blake3(data, len(data), out.ctypes.data)
    return out.tobytes()


def sha256_witness(data: bytes, timestamp: int = 0) -> dict:
    """sha256_hash_with_witness (crypto.rs:223-297) minus memory traffic; raises for len >= 56."""
    out = np.zeros(608, dtype="<u4")
    rc = lib().zo_sha256_witness(data, len(data), timestamp, out.ctypes.data)
    if rc != 0:
        raise OracleError(rc, "SHA-256 witness collection only supports messages < 56 bytes")
    return {"message_block": out[0:16].copy(), "initial_state": out[16:24].copy(), "message_schedule": out[24:88].copy(),
            "round_states": out[88:600].reshape(64, 8).copy(), "final_state": out[600:608].copy(), "flat": out,
            "timestamp": timestamp}


def decode(word: int):
    f = [C.c_uint8(), C.c_uint8(), C.c_uint8(), C.c_uint8(), C.c_int32(), C.c_uint8()]
    rc = lib().zo_decode(word, *[C.addressof(x) for x in f])
    if rc:
        return None
    return dict(op=f[0].value, rd=f[1].value, rs1=f[2].value, rs2=f[3].value, imm=f[4].value, shamt=f[5].value)


def value40(op: int, a: int, b: int) -> int:
    out = C.c_uint64()
    lib().zo_value40_op(op, a & (2**64 - 1), b & (2**64 - 1), C.byref(out))
    return out.value
