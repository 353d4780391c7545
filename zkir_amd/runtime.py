"""Host-side mirror of the reference's `zkir_runtime` API on top of the C ABI (include/zkir_amd.h).

    reference (Rust)                                   here
    -----------------------------------------------    ---------------------------------------------
    VMConfig {max_cycles, trace, enable_* ..}          VMConfig                      vm.rs:15-50
    VM::new(program, inputs, config)                   VM(program, inputs, config)   vm.rs:138
    VM::run(self) -> Result<ExecutionResult>           VM.run() -> ExecutionResult   vm.rs:208
    ExecutionResult.{cycles, outputs, halt_reason,     same attribute names          vm.rs:54-78
       range_check_witnesses, execution_trace,
       normalization_witnesses}
    ExecutionResult::get_memory_trace()                ExecutionResult.get_memory_trace()   vm.rs:85-94
    RuntimeError::{MisalignedAccess, ..}               RuntimeError(code, message)   error.rs:7-37
    zkir_runtime::run(program, inputs)                 run(program, inputs)          lib.rs:59-62

The wide execution trace stays resident in HBM (SoA columns); `execution_trace.column(...)` copies one
column to the host, `execution_trace.rows()` reassembles reference-shaped TraceRow records (for tests).
There is no CPU fallback: without the built HIP library or without a GPU, trace collection raises.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .spec import Program

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZKIR_AMD_LIB: another build of the same library (kernel experiments: scripts/build_variant.py); the default is the in-tree build
_SO = os.environ.get("ZKIR_AMD_LIB") or os.path.join(_HERE, "libzkir_amd.so")

(ZKIR_OK, ERR_MISALIGNED, ERR_INVALID_MEMORY, ERR_DIV_ZERO, ERR_INVALID_SYSCALL, ERR_DECODE, ERR_OTHER, ERR_BAD_PROGRAM,
 ERR_DEVICE, ERR_ARGUMENT) = range(10)
HALT_EBREAK, HALT_EXIT, HALT_CYCLE_LIMIT = 0, 1, 2

REG_EVENT_DTYPE = np.dtype([("value", "<u8"), ("payload", "<u8"), ("max_bits", "<u4"), ("vis", "<u4"), ("reg", "u1"),
                            ("state", "u1"), ("tag", "u1"), ("pad", "u1", (5,))])
MEM_EVENT_DTYPE = np.dtype([("address", "<u8"), ("value", "<u8"), ("row", "<u4"), ("is_write", "u1"), ("width", "u1"), ("pad", "<u2")])
RC_EVENT_DTYPE = np.dtype([("value", "<u8"), ("pc", "<u8")])
NORM_EVENT_DTYPE = np.dtype([("cycle", "<u8"), ("pc", "<u8"), ("raw_value", "<u8"), ("reg", "u1"), ("state", "u1"), ("opcode", "u1"),
                             ("pad", "u1", (5,))])
SHA_BLOCK_DTYPE = np.dtype([("message_block", "<u4", (16,)), ("timestamp", "<u8")])
assert REG_EVENT_DTYPE.itemsize == 32 and MEM_EVENT_DTYPE.itemsize == 24 and NORM_EVENT_DTYPE.itemsize == 32 and SHA_BLOCK_DTYPE.itemsize == 72

_MEMOP_DTYPE = np.dtype([("address", "<u8"), ("value", "<u8"), ("timestamp", "<u8"), ("is_write", "u1"), ("width", "u1"),
                         ("bound_bits", "<u4"), ("bound_tag", "u1"), ("bound_payload", "<u8")])
_NORM_DTYPE = np.dtype([("cycle", "<u8"), ("pc", "<u8"), ("reg", "u1"), ("accumulated", "<u8", (2,)), ("normalized", "<u4", (2,)),
                        ("carries", "<u4", (2,)), ("normalized_bits", "u1"), ("limb_bits", "u1"), ("cause", "u1"), ("opcode", "u1")])

FIELD_CYCLE, FIELD_PC, FIELD_INSTRUCTION, FIELD_REGISTERS, FIELD_BOUND_BITS, FIELD_BOUND_TAG, FIELD_BOUND_PAYLOAD, FIELD_REG_STATE = range(8)
_FIELD_DTYPE = {0: "<u8", 1: "<u8", 2: "<u4", 3: "<u8", 4: "<u4", 5: "u1", 6: "<u8", 7: "u1"}


class RuntimeError(Exception):  # noqa: A001 - mirrors zkir_runtime::RuntimeError
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code, self.message = code, message


class VmConfigC(C.Structure):
    _fields_ = [("max_cycles", C.c_uint64), ("trace", C.c_uint8), ("enable_range_checking", C.c_uint8),
                ("enable_execution_trace", C.c_uint8), ("enable_deferred_model", C.c_uint8)]


class TraceColumnsC(C.Structure):
    _fields_ = [("cycle", C.c_void_p), ("pc", C.c_void_p), ("instruction", C.c_void_p), ("registers", C.c_void_p),
                ("bound_bits", C.c_void_p), ("bound_tag", C.c_void_p), ("bound_payload", C.c_void_p), ("reg_state", C.c_void_p),
                ("reg_stride", C.c_uint64)]


class TraceFillArgsC(C.Structure):
    _fields_ = [("events", C.c_void_p), ("tile_ev_off", C.c_void_p), ("tile_snap", C.c_void_p), ("n_rows", C.c_uint64),
                ("cycle_base", C.c_uint64), ("tile_rows", C.c_uint32), ("n_events", C.c_uint32), ("out", TraceColumnsC)]


class MemopColumnsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("address", "value", "timestamp", "is_write", "width", "bound_bits", "bound_tag", "bound_payload")]


class NormColumnsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("cycle", "pc", "reg", "opcode", "accumulated0", "accumulated1", "normalized0", "normalized1",
                                         "carry0", "carry1")]


class ProverParamsC(C.Structure):             # zkir_prover_params
    _fields_ = [("mode", C.c_uint32), ("num_queries", C.c_uint32), ("pow_bits", C.c_uint32)]


class PublicInputsC(C.Structure):             # zkir_public_inputs
    _fields_ = [("n_real", C.c_uint64), ("entry_point", C.c_uint64), ("deferred", C.c_uint32), ("fri_params", C.c_uint32),
                ("program_digest", C.c_uint32 * 4), ("io_digest", C.c_uint32 * 4),
                ("program_blob", C.c_void_p), ("program_blob_len", C.c_uint64),      # borrowed pointer (prover side): see with_program()
                # mode 2 (`deferred` == 2: default VM mode + the I/O argument): the tapes and the halt reason in the clear (borrowed pointers: with_io()); for a SEGMENT
                # the WRITE / READ ecalls executed before its first row
                ("inputs", C.c_void_p), ("n_inputs", C.c_uint64), ("outputs", C.c_void_p), ("n_outputs", C.c_uint64), ("halt_kind", C.c_uint32), ("reserved2", C.c_uint32),
                ("halt_code", C.c_uint64), ("writes_before", C.c_uint64), ("reads_before", C.c_uint64),
                # mode 3 (`deferred` == 3: mode 2 + the memory argument): the memory witness of the run (borrowed pointers into a MemcheckWitness: with_memory())
                ("mem_old", C.c_void_p), ("mem_told", C.c_void_p), ("cell_addr", C.c_void_p), ("cell_bytes", C.c_void_p), ("cell_time", C.c_void_p), ("n_cells", C.c_uint64),
                # mode 4 (`deferred` == 4: mode 3 + the wide-arithmetic class + hash syscalls as a tape): the hash calls as the proof's hash section (borrowed from a MemcheckWitness)
                ("hash_section", C.c_void_p), ("hash_section_words", C.c_uint64)]

    def with_params(self, num_queries: int = 0, pow_bits: int = 0) -> "PublicInputsC":
        """zkir_public_inputs_set_params: the prover's FRI parameters (0 = the defaults: 50 queries, 12 grinding bits; accepted 50..128 / 12..24).  A verifier given this
        struct as `expect` requires exactly them."""
        pr = ProverParamsC(int(self.deferred), int(num_queries), int(pow_bits))
        L = lib()
        L.zkir_public_inputs_set_params.restype = C.c_int
        L.zkir_public_inputs_set_params.argtypes = [C.c_void_p, C.c_void_p]
        rc = L.zkir_public_inputs_set_params(C.byref(self), C.byref(pr))
        if rc != ZKIR_OK:
            _raise(rc)
        return self

    def with_memory(self, witness: "MemcheckWitness") -> "PublicInputsC":
        """zkir_public_inputs_set_memory: mode 3 — point the struct at the run's memory witness (kept alive by the struct)."""
        self._mem_ref = witness
        lib().zkir_public_inputs_set_memory(C.byref(self), witness._h)
        return self

    def with_io(self, inputs, outputs, halt=None, writes_before=None, reads_before=None) -> "PublicInputsC":
        """Point the struct at its own copies of the I/O tapes (kept alive by the struct); halt = a HaltReason or (kind, code)."""
        self._in_ref = np.ascontiguousarray(np.asarray([int(x) & (2**64 - 1) for x in inputs], dtype=np.uint64))
        self._out_ref = np.ascontiguousarray(np.asarray([int(x) & (2**64 - 1) for x in outputs], dtype=np.uint64))
        self.inputs, self.n_inputs = (self._in_ref.ctypes.data if len(self._in_ref) else None), len(self._in_ref)
        self.outputs, self.n_outputs = (self._out_ref.ctypes.data if len(self._out_ref) else None), len(self._out_ref)
        if halt is not None:
            kind, code = (halt.kind, halt.code) if hasattr(halt, "kind") else halt
            self.halt_kind, self.halt_code = int(kind), int(code or 0) if int(kind) == 1 else 0
        if writes_before is not None:
            self.writes_before = int(writes_before)
        if reads_before is not None:
            self.reads_before = int(reads_before)
        return self

    def with_program(self, blob: bytes) -> "PublicInputsC":
        """Point the struct at `blob` (kept alive by the struct): needed after the struct has been copied byte-wise or sent to another
        process, where the borrowed pointer means nothing."""
        self._blob_ref = bytes(blob)
        self._blob_buf = C.create_string_buffer(self._blob_ref, len(self._blob_ref))
        self.program_blob = C.cast(self._blob_buf, C.c_void_p).value
        self.program_blob_len = len(self._blob_ref)
        return self

    def copy(self) -> "PublicInputsC":
        q = PublicInputsC.from_buffer_copy(bytes(self))
        q.with_io(getattr(self, "_in_ref", []), getattr(self, "_out_ref", []))
        if hasattr(self, "_mem_ref"):
            q.with_memory(self._mem_ref)
        return q.with_program(getattr(self, "_blob_ref", b""))


class MemcheckWitness:
    """zkir_memcheck_witness_of: the memory witness of a whole run (mode 3) — per row the accessed 8-byte cell's bytes before the access and the time of its previous
    access, and the touched cells; a sequential host replay (memory is a chain).  Refused for shards / windows, addresses of 2^40 or more, executed hash syscalls."""

    def __init__(self, log: "DeltaLog", program, mode: int = 3):
        """mode 4 (zkir_memcheck_witness_of_mode): the run's hash syscalls are replayed too and recorded as the proof's hash section (n_hash_calls of them)."""
        blob = bytes(program) if isinstance(program, (bytes, bytearray)) else program.to_bytes()
        L = lib()
        L.zkir_memcheck_witness_of_mode.restype = C.c_int
        L.zkir_memcheck_witness_of_mode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.zkir_memcheck_witness_n_hash_calls.restype = C.c_uint64; L.zkir_memcheck_witness_n_hash_calls.argtypes = [C.c_void_p]
        L.zkir_memcheck_witness_free.restype = None; L.zkir_memcheck_witness_free.argtypes = [C.c_void_p]
        L.zkir_memcheck_witness_n_cells.restype = C.c_uint64; L.zkir_memcheck_witness_n_cells.argtypes = [C.c_void_p]
        L.zkir_memcheck_witness_n_accesses.restype = C.c_uint64; L.zkir_memcheck_witness_n_accesses.argtypes = [C.c_void_p]
        L.zkir_public_inputs_set_memory.restype = None; L.zkir_public_inputs_set_memory.argtypes = [C.c_void_p, C.c_void_p]
        h = C.c_void_p()
        rc = L.zkir_memcheck_witness_of_mode(log._h, blob, len(blob), int(mode), C.byref(h))
        if rc != ZKIR_OK:
            _raise(rc)
        self._h = h
        self.n_hash_calls = int(L.zkir_memcheck_witness_n_hash_calls(h))
        self.n_cells = int(L.zkir_memcheck_witness_n_cells(h))
        self.n_accesses = int(L.zkir_memcheck_witness_n_accesses(h))

    def __del__(self):
        try:
            if self._h:
                lib().zkir_memcheck_witness_free(self._h)
                self._h = None
        except Exception:
            pass


class MemoryWitnessC(C.Structure):            # zkir_memory_witness
    _fields_ = [("n_ops", C.c_uint64), ("n_rows", C.c_uint64), ("row_order", MemopColumnsC), ("row_offsets", C.c_void_p), ("sorted", MemopColumnsC)]


class RangeCheckWitnessC(C.Structure):        # zkir_range_check_witness
    _fields_ = [("n_checks", C.c_uint64), ("n_witnesses", C.c_uint64), ("witness_offsets", C.c_void_p), ("witness_cycles", C.c_void_p),
                ("value", C.c_void_p), ("pc", C.c_void_p), ("chunks", C.c_void_p), ("chunk_stride", C.c_uint64), ("chunk_bits", C.c_uint32),
                ("multiplicity", C.c_void_p)]


class NormalizationWitnessC(C.Structure):     # zkir_normalization_witness
    _fields_ = [("n_events", C.c_uint64), ("columns", NormColumnsC)]


class Sha256WitnessC(C.Structure):            # zkir_sha256_witness
    _fields_ = [("n_blocks", C.c_uint64), ("columns", C.c_void_p), ("stride", C.c_uint64), ("timestamps", C.c_void_p)]


_lib = None


def lib() -> C.CDLL:
    """Load libzkir_amd.so; fails loudly if it has not been built (python -m zkir_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(f"{_SO} is missing: build it with `python zkir_amd/build.py` (hipcc --offload-arch=gfx950). "
                          "zkir_amd has no pure-Python or CPU fallback.")
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so (soname libamdhip64.so.7, same as
    # /opt/rocm's).  Importing torch first makes the loader resolve our DT_NEEDED to that already-loaded copy;
    # loading ours first would give the process two runtimes, and whichever touches the device second fails.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_SO)
    L.zkir_last_error.restype = C.c_char_p
    L.zkir_version.restype = C.c_char_p
    L.zkir_interpret.restype = C.c_int
    L.zkir_interpret.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(VmConfigC), C.c_uint32,
                                 C.POINTER(C.c_void_p)]
    L.zkir_delta_log_free.argtypes = [C.c_void_p]
    L.zkir_delta_log_free.restype = None
    for name, res in [("cycles", C.c_uint64), ("halt_kind", C.c_int), ("halt_code", C.c_uint64), ("n_outputs", C.c_size_t),
                      ("outputs", C.c_void_p), ("n_rows", C.c_uint64), ("tile_rows", C.c_uint32), ("pc", C.c_void_p),
                      ("inst", C.c_void_p), ("n_reg_events", C.c_size_t), ("reg_events", C.c_void_p), ("n_tiles", C.c_size_t),
                      ("tile_ev_off", C.c_void_p), ("tile_snap", C.c_void_p), ("n_mem_events", C.c_size_t),
                      ("mem_events", C.c_void_p), ("n_rc_events", C.c_size_t), ("rc_events", C.c_void_p),
                      ("n_rc_witnesses", C.c_size_t), ("rc_offsets", C.c_void_p), ("rc_cycles", C.c_void_p), ("rc_chunk_bits", C.c_uint32),
                      ("n_norm_events", C.c_size_t), ("norm_events", C.c_void_p), ("n_sha_blocks", C.c_size_t),
                      ("sha_blocks", C.c_void_p)]:
        f = getattr(L, "zkir_delta_log_" + name)
        f.restype = res
        f.argtypes = [C.c_void_p]
    L.zkir_delta_log_shard.restype = C.c_int
    L.zkir_delta_log_shard.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
    L.zkir_delta_log_cycle_base.restype = C.c_uint64
    L.zkir_delta_log_cycle_base.argtypes = [C.c_void_p]
    L.zkir_delta_log_window_open.restype = C.c_int
    L.zkir_delta_log_window_open.argtypes = [C.c_void_p]
    L.zkir_interpret_window.restype = C.c_int
    L.zkir_interpret_window.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(VmConfigC), C.c_uint32, C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_void_p)]
    L.zkir_exec_window.restype = C.c_int
    L.zkir_exec_window.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(VmConfigC), C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
    L.zkir_exec_shard.restype = C.c_int
    L.zkir_exec_shard.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
    L.zkir_trace_fill_launch.restype = C.c_int
    L.zkir_trace_fill_launch.argtypes = [C.POINTER(TraceFillArgsC), C.c_void_p]
    L.zkir_trace_fill_bytes.restype = C.c_uint64
    L.zkir_trace_fill_bytes.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    V, U64, U32 = C.c_void_p, C.c_uint64, C.c_uint32
    for name, args in [("zkir_memops_expand_launch", [V, U64, U64, C.POINTER(MemopColumnsC), V]),
                       ("zkir_memops_row_offsets_launch", [V, U64, U64, V, V]),
                       ("zkir_memops_expand_csr_launch", [V, U64, U64, U64, C.POINTER(MemopColumnsC), V, V, V]),
                       ("zkir_memops_sort_prepared_launch", [V, U64, U64, V, V, C.POINTER(MemopColumnsC), V]),
                       ("zkir_memops_sort_launch", [V, U64, U64, U64, V, V, C.POINTER(MemopColumnsC), V]),
                       ("zkir_range_check_expand_launch", [V, U64, U32, V, V, V, U64, V, V]),
                       ("zkir_norm_expand_launch", [V, U64, C.POINTER(NormColumnsC), V]),
                       ("zkir_sha256_chip_launch", [V, U64, V, U64, V, V])]:
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = args
    L.zkir_stark_ctx_create.restype = C.c_int
    L.zkir_stark_ctx_create.argtypes = [U32, U32, C.POINTER(V)]
    L.zkir_stark_ctx_free.restype = None
    L.zkir_stark_ctx_free.argtypes = [V]
    L.zkir_main_trace_width.restype = U32
    if hasattr(L, "zkir_main_trace_width_for"):               # absent from older builds loaded through ZKIR_AMD_LIB (kernel experiments)
        L.zkir_main_trace_width_for.restype = U32; L.zkir_main_trace_width_for.argtypes = [U32]
    L.zkir_modmul_peak_per_s.restype = C.c_double
    L.zkir_modmul_peak_per_s.argtypes = [V]
    L.zkir_padded_log_n.restype = U32
    L.zkir_padded_log_n.argtypes = [U64]
    for name, args in [("zkir_main_trace_launch", [C.POINTER(TraceColumnsC), U64, U32, V, V]), ("zkir_lde_launch", [V, V, U32, V, V]),
                       ("zkir_merkle_commit_launch", [V, V, U32, U64, V, V]), ("zkir_merkle_leaves_launch", [V, V, U32, U64, V, V]),
                       ("zkir_merkle_cap_launch", [V, V, U64, V])]:
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = args
    L.zkir_prove.restype = C.c_int
    L.zkir_prove.argtypes = [V, C.POINTER(TraceColumnsC), C.POINTER(PublicInputsC), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(U64), C.POINTER(C.c_float), V]
    L.zkir_verify.restype = C.c_int
    L.zkir_verify.argtypes = [V, U64, C.POINTER(PublicInputsC)]
    L.zkir_verify_segment.restype = C.c_int
    L.zkir_verify_segment.argtypes = [V, U64, C.POINTER(PublicInputsC), V, V]
    L.zkir_verify_chain.restype = C.c_int
    L.zkir_verify_chain.argtypes = [V, V, C.c_uint32, C.POINTER(PublicInputsC)]
    L.zkir_proof_state_words.restype = C.c_uint32
    L.zkir_digest_bytes.restype = None
    L.zkir_digest_bytes.argtypes = [C.c_char_p, C.c_size_t, V]
    L.zkir_public_inputs_of.restype = C.c_int
    L.zkir_public_inputs_of.argtypes = [V, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, U32, C.POINTER(PublicInputsC)]
    L.zkir_proof_version.restype = U32
    L.zkir_proof_free.restype = None
    L.zkir_proof_free.argtypes = [C.POINTER(C.c_uint32)]
    L.zkir_proof_num_queries.restype = U32
    L.zkir_poseidon2_permute.restype = None; L.zkir_poseidon2_permute.argtypes = [C.c_void_p]
    L.zkir_poseidon2_permute_scaled.restype = None; L.zkir_poseidon2_permute_scaled.argtypes = [C.c_void_p, C.c_uint32]
    L.zkir_exec.restype = C.c_int
    L.zkir_exec.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(VmConfigC), C.POINTER(C.c_void_p)]
    L.zkir_result_free.argtypes = [C.c_void_p]
    L.zkir_result_free.restype = None
    L.zkir_result_delta_log.restype = C.c_void_p
    L.zkir_result_delta_log.argtypes = [C.c_void_p]
    L.zkir_result_trace.restype = C.POINTER(TraceColumnsC)
    L.zkir_result_trace.argtypes = [C.c_void_p]
    L.zkir_result_stage_ms.restype = None
    L.zkir_result_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.zkir_result_copy_column.restype = C.c_int
    L.zkir_result_copy_column.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for name, st in [("memory_trace", MemoryWitnessC), ("range_check_witnesses", RangeCheckWitnessC),
                     ("normalization_witnesses", NormalizationWitnessC), ("sha256_witnesses", Sha256WitnessC)]:
        f = getattr(L, "zkir_result_" + name)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.POINTER(st)]
    L.zkir_host_to_device.restype = C.c_int
    L.zkir_host_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.zkir_device_to_host.restype = C.c_int
    L.zkir_device_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _lib = L
    return L


def _raise(code: int):
    raise RuntimeError(code, lib().zkir_last_error().decode())


def _view(ptr, n, dtype) -> np.ndarray:
    dtype = np.dtype(dtype)
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


@dataclass
class VMConfig:
    """vm.rs:15-50 (same defaults)."""
    max_cycles: int = 1_000_000
    trace: bool = False
    enable_range_checking: bool = False
    enable_execution_trace: bool = False
    enable_deferred_model: bool = False

    def _c(self) -> VmConfigC:
        return VmConfigC(self.max_cycles, int(self.trace), int(self.enable_range_checking), int(self.enable_execution_trace),
                         int(self.enable_deferred_model))


@dataclass(frozen=True)
class HaltReason:
    """state.rs:8-15."""
    kind: int
    code: int = 0

    @staticmethod
    def Ebreak(): return HaltReason(HALT_EBREAK)
    @staticmethod
    def Exit(code: int): return HaltReason(HALT_EXIT, code)
    @staticmethod
    def CycleLimit(): return HaltReason(HALT_CYCLE_LIMIT)


class DeltaLog:
    """Owner of a zkir_delta_log handle with zero-copy numpy views (valid while this object lives)."""

    def __init__(self, handle: int, owned: bool = True):
        self._h, self._owned = handle, owned
        self._uploads = []                                    # events of asynchronous copies out of this log's (pinned) buffers: close() waits for them (pipeline.upload)
        L = lib()
        h = handle
        self.cycles = L.zkir_delta_log_cycles(h)
        self.halt_reason = HaltReason(L.zkir_delta_log_halt_kind(h), L.zkir_delta_log_halt_code(h) if L.zkir_delta_log_halt_kind(h) == HALT_EXIT else 0)
        self.outputs = _view(L.zkir_delta_log_outputs(h), L.zkir_delta_log_n_outputs(h), "<u8").tolist()
        self.n_rows = L.zkir_delta_log_n_rows(h)
        self.cycle_base = L.zkir_delta_log_cycle_base(h)
        self.window_open = bool(L.zkir_delta_log_window_open(h))      # a trace window that ended before the run did
        self.tile_rows = L.zkir_delta_log_tile_rows(h)
        self.pc = _view(L.zkir_delta_log_pc(h), self.n_rows, "<u8")
        self.inst = _view(L.zkir_delta_log_inst(h), self.n_rows, "<u4")
        self.reg_events = _view(L.zkir_delta_log_reg_events(h), L.zkir_delta_log_n_reg_events(h), REG_EVENT_DTYPE)
        self.n_tiles = L.zkir_delta_log_n_tiles(h)
        self.tile_ev_off = _view(L.zkir_delta_log_tile_ev_off(h), self.n_tiles + 1 if self.n_rows else 0, "<u4")
        self.tile_snap = _view(L.zkir_delta_log_tile_snap(h), self.n_tiles * 16, "<u4").reshape(-1, 16)
        self.mem_events = _view(L.zkir_delta_log_mem_events(h), L.zkir_delta_log_n_mem_events(h), MEM_EVENT_DTYPE)
        self.rc_events = _view(L.zkir_delta_log_rc_events(h), L.zkir_delta_log_n_rc_events(h), RC_EVENT_DTYPE)
        self.rc_offsets = _view(L.zkir_delta_log_rc_offsets(h), L.zkir_delta_log_n_rc_witnesses(h) + 1, "<u8")
        self.rc_cycles = _view(L.zkir_delta_log_rc_cycles(h), L.zkir_delta_log_n_rc_witnesses(h), "<u8")
        self.rc_chunk_bits = L.zkir_delta_log_rc_chunk_bits(h)
        self.norm_events = _view(L.zkir_delta_log_norm_events(h), L.zkir_delta_log_n_norm_events(h), NORM_EVENT_DTYPE)
        self.sha_blocks = _view(L.zkir_delta_log_sha_blocks(h), L.zkir_delta_log_n_sha_blocks(h), SHA_BLOCK_DTYPE)

    def shard(self, row_begin: int, row_end: int) -> "DeltaLog":
        """zkir_delta_log_shard: self-contained delta log of rows [row_begin, row_end) (multi-GPU row sharding)."""
        out = C.c_void_p()
        rc = lib().zkir_delta_log_shard(self._h, row_begin, row_end, C.byref(out))
        if rc != ZKIR_OK:
            _raise(rc)
        return DeltaLog(out.value)

    def close(self):
        for ev in self._uploads:
            ev.synchronize()
        self._uploads = []
        if self._owned and self._h:
            lib().zkir_delta_log_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def public_inputs(log: DeltaLog, program: Program | bytes, inputs: Sequence[int] = (), deferred: bool = False, io_mode: bool = False, mem_mode: bool = False,
                  mem_witness: str = "device", num_queries: int = 0, pow_bits: int = 0, wide_mode: bool = False) -> PublicInputsC:
    """zkir_public_inputs_of: what a proof of this run is bound to (row count, mode, entry pc, program digest, io digest).  io_mode = mode 2: the default VM mode
    with the I/O argument (WRITE / READ ecalls tied to the tapes, which the proof then carries).  mem_mode = mode 3: mode 2 with the memory argument (loads and stores
    constrained, every access tied to a consistent memory; the proof carries the touched cells).  mem_witness = "device": zkir_prove computes the run's memory witness on the
    GPU (memcheck.hip: address-major sort + segmented scan); "host": it is computed here by the host's sequential replay (zkir_memcheck_witness_of — the independent
    implementation; needs no device) and handed to zkir_prove.  num_queries / pow_bits: the prover's FRI parameters (zkir_prover_params; 0 = the defaults, 50 + 12).
    wide_mode = mode 4 (round 6): mode 3 with MULH / DIVU / REMU / DIV / REM constrained — by a chunk relation on operands below 2^40, through the wide tape (a record per row, the verifier
    recomputes the result on the raw 64-bit registers) above — hash syscalls as a tape, the code segment's boundary cell: every run of the VM that stays below 2^40 in its addresses and off
    its own code has a mode-4 proof."""
    blob = bytes(program) if isinstance(program, (bytes, bytearray)) else program.to_bytes()
    arr = (C.c_uint64 * max(1, len(inputs)))(*[int(x) & (2**64 - 1) for x in inputs])
    out = PublicInputsC()
    assert not (deferred and (io_mode or mem_mode or wide_mode)), "the I/O and memory arguments are stated for the default VM mode"
    rc = lib().zkir_public_inputs_of(log._h, blob, len(blob), arr, len(inputs), 4 if wide_mode else 3 if mem_mode else 2 if io_mode else int(deferred), C.byref(out))
    if rc != ZKIR_OK:
        _raise(rc)
    out.with_io(list(inputs), list(log.outputs))      # the C call borrowed temporaries: re-point at arrays / bytes this struct owns
    if wide_mode and mem_witness == "device" and log.n_rows and int(np.count_nonzero((log.inst & 0x7F) == 0x50)) > (1 if log.halt_reason.kind == HALT_EXIT else 0):
        mem_witness = "host"                          # a run that may make hash syscalls (an ECALL beside the exit) is proven from the host witness: the hash tape comes with it
    if (mem_mode or wide_mode) and mem_witness == "host":
        out.with_memory(MemcheckWitness(log, blob, 4 if wide_mode else 3))
    if num_queries or pow_bits:
        out.with_params(num_queries, pow_bits)
    return out.with_program(blob)


def verify_segment(proof: np.ndarray, expect: Optional[PublicInputsC] = None):
    """zkir_verify_segment: a SEGMENT of a run -> (code, first_state u32[68], last_state u32[68])."""
    proof = np.ascontiguousarray(proof, dtype=np.uint32)
    first, last = np.zeros(68, np.uint32), np.zeros(68, np.uint32)
    rc = lib().zkir_verify_segment(proof.ctypes.data, len(proof), C.byref(expect) if expect is not None else None, first.ctypes.data, last.ctypes.data)
    return rc, first, last


def verify_chain(proofs, expect: Optional[PublicInputsC] = None) -> int:
    """zkir_verify_chain: the segments of one run, in order; expect.n_real = the run's total executed rows."""
    ps = [np.ascontiguousarray(p, dtype=np.uint32) for p in proofs]
    ptrs = (C.c_void_p * len(ps))(*[p.ctypes.data for p in ps])
    lens = (C.c_uint64 * len(ps))(*[len(p) for p in ps])
    return lib().zkir_verify_chain(ptrs, lens, len(ps), C.byref(expect) if expect is not None else None)


def _u64s(a):
    return np.ascontiguousarray(np.asarray([int(x) & (2**64 - 1) for x in a], dtype=np.uint64))


def verify_io(proof: np.ndarray, expect: Optional[PublicInputsC], inputs, outputs, halt) -> int:
    """zkir_verify_io: zkir_verify + the run's claim in the clear — `inputs`, `outputs` and `halt` (a HaltReason or (kind, code)) must hash to the proof's io digest (50) and
    the halt row must be the instruction the halt reason names (52: not that instruction; 53: not that exit code)."""
    proof, i, o = np.ascontiguousarray(proof, dtype=np.uint32), _u64s(inputs), _u64s(outputs)
    kind, code = (halt.kind, halt.code) if hasattr(halt, "kind") else halt
    L = lib()
    L.zkir_verify_io.restype = C.c_int
    L.zkir_verify_io.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint64]
    return L.zkir_verify_io(proof.ctypes.data, len(proof), C.byref(expect) if expect is not None else None, i.ctypes.data, len(i), o.ctypes.data, len(o), int(kind), int(code or 0))


def verify_chain_io(proofs, expect: Optional[PublicInputsC], inputs, outputs, halt) -> int:
    """zkir_verify_chain_io: the same for a run proven in segments."""
    ps = [np.ascontiguousarray(p, dtype=np.uint32) for p in proofs]
    ptrs = (C.c_void_p * len(ps))(*[p.ctypes.data for p in ps])
    lens = (C.c_uint64 * len(ps))(*[len(p) for p in ps])
    i, o = _u64s(inputs), _u64s(outputs)
    kind, code = (halt.kind, halt.code) if hasattr(halt, "kind") else halt
    L = lib()
    L.zkir_verify_chain_io.restype = C.c_int
    L.zkir_verify_chain_io.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint64]
    return L.zkir_verify_chain_io(ptrs, lens, len(ps), C.byref(expect) if expect is not None else None, i.ctypes.data, len(i), o.ctypes.data, len(o), int(kind), int(code or 0))


def verify(proof: np.ndarray, expect: Optional[PublicInputsC] = None) -> int:
    """zkir_verify (host only): 0 = accepted, otherwise the number of the failed check."""
    proof = np.ascontiguousarray(proof, dtype=np.uint32)
    return lib().zkir_verify(proof.ctypes.data, len(proof), C.byref(expect) if expect is not None else None)


def interpret(program: Program | bytes, inputs: Sequence[int] = (), config: Optional[VMConfig] = None, tile_rows: int = 0,
              window: Optional[Tuple[int, int]] = None) -> DeltaLog:
    """Host stage only (zkir_interpret): run the program, return the delta log.  Never touches a GPU.
    window = (row_begin, row_end): zkir_interpret_window — rows before row_begin are executed untraced, the log holds the window."""
    blob = program if isinstance(program, (bytes, bytearray)) else program.to_bytes()
    cfg = (config or VMConfig())._c()
    arr = (C.c_uint64 * max(1, len(inputs)))(*[int(x) & (2**64 - 1) for x in inputs])
    out = C.c_void_p()
    if window is None:
        rc = lib().zkir_interpret(bytes(blob), len(blob), arr, len(inputs), C.byref(cfg), tile_rows, C.byref(out))
    else:
        rc = lib().zkir_interpret_window(bytes(blob), len(blob), arr, len(inputs), C.byref(cfg), tile_rows, int(window[0]), int(window[1]), C.byref(out))
    if rc != ZKIR_OK:
        _raise(rc)
    return DeltaLog(out.value)


class ExecutionTrace:
    """Device-resident SoA execution trace (TraceRow schema, trace.rs:24-50)."""

    def __init__(self, result_handle: int, n_rows: int):
        self._r, self.n_rows = result_handle, n_rows
        self.columns = lib().zkir_result_trace(result_handle).contents if n_rows else None

    def __len__(self):
        return self.n_rows

    def column(self, field: int, reg: int = 0) -> np.ndarray:
        out = np.empty(self.n_rows, dtype=_FIELD_DTYPE[field])
        if self.n_rows:
            rc = lib().zkir_result_copy_column(self._r, field, reg, out.ctypes.data)
            if rc != ZKIR_OK:
                _raise(rc)
        return out

    _ROW_DTYPE = np.dtype([("cycle", "<u8"), ("pc", "<u8"), ("instruction", "<u4"), ("registers", "<u8", (16,)),
                           ("bound_bits", "<u4", (16,)), ("bound_tag", "u1", (16,)), ("bound_payload", "<u8", (16,)), ("reg_state", "u1", (16,))])

    def rows_window(self, lo: int, hi: int) -> np.ndarray:
        """Rows [lo, hi) as packed reference-shaped records, copied straight from the device columns (sampled parity checks at
        sizes where copying whole columns is wasteful)."""
        assert 0 <= lo <= hi <= self.n_rows
        n = hi - lo
        out = np.zeros(n, dtype=self._ROW_DTYPE)
        if n == 0:
            return out
        c, L = self.columns, lib()

        def grab(base, elt, dtype, reg=None):
            buf = np.empty(n, dtype=dtype)
            off = (lo if reg is None else reg * c.reg_stride + lo) * elt
            rc = L.zkir_device_to_host(buf.ctypes.data, base + off, n * elt)
            if rc != ZKIR_OK:
                _raise(rc)
            return buf
        out["cycle"] = grab(c.cycle, 8, "<u8"); out["pc"] = grab(c.pc, 8, "<u8"); out["instruction"] = grab(c.instruction, 4, "<u4")
        for r in range(16):
            out["registers"][:, r] = grab(c.registers, 8, "<u8", r)
            out["bound_bits"][:, r] = grab(c.bound_bits, 4, "<u4", r)
            out["bound_tag"][:, r] = grab(c.bound_tag, 1, "u1", r)
            out["bound_payload"][:, r] = grab(c.bound_payload, 8, "<u8", r)
            out["reg_state"][:, r] = grab(c.reg_state, 1, "u1", r)
        return out

    def rows(self) -> np.ndarray:
        """All rows as packed reference-shaped records (same dtype as the test oracle's rows)."""
        dt = self._ROW_DTYPE
        out = np.zeros(self.n_rows, dtype=dt)
        out["cycle"] = self.column(FIELD_CYCLE)
        out["pc"] = self.column(FIELD_PC)
        out["instruction"] = self.column(FIELD_INSTRUCTION)
        for r in range(16):
            out["registers"][:, r] = self.column(FIELD_REGISTERS, r)
            out["bound_bits"][:, r] = self.column(FIELD_BOUND_BITS, r)
            out["bound_tag"][:, r] = self.column(FIELD_BOUND_TAG, r)
            out["bound_payload"][:, r] = self.column(FIELD_BOUND_PAYLOAD, r)
            out["reg_state"][:, r] = self.column(FIELD_REG_STATE, r)
        return out


class ExecutionResult:
    """vm.rs:54-78."""

    def __init__(self, result_handle: Optional[int], log: DeltaLog, blob: bytes = b"", inputs: Sequence[int] = (), config: Optional[VMConfig] = None):
        self._r = result_handle
        self._log = log
        self._blob, self._inputs, self._config = blob, list(inputs), config or VMConfig()
        self.cycles: int = log.cycles
        self.outputs: List[int] = log.outputs
        self.halt_reason: HaltReason = log.halt_reason
        self.execution_trace = ExecutionTrace(result_handle, log.n_rows) if result_handle else ExecutionTrace.__new__(ExecutionTrace)
        if not result_handle:
            self.execution_trace._r, self.execution_trace.n_rows, self.execution_trace.columns = None, 0, None
        self.delta_log = log

    # -- the remaining ExecutionResult members (vm.rs:64-103): device columns behind the C handle (zkir_result_*), copied to the
    #    host here as reference-shaped records for the tests --
    def _d2h(self, ptr, n, dtype) -> np.ndarray:
        out = np.empty(n, dtype=dtype)
        if n:
            rc = lib().zkir_device_to_host(out.ctypes.data, ptr, out.nbytes)
            if rc != ZKIR_OK:
                _raise(rc)
        return out

    def _memops(self, cols: MemopColumnsC, n: int) -> np.ndarray:
        out = np.zeros(n, dtype=_MEMOP_DTYPE)
        for name in _MEMOP_DTYPE.names:
            out[name] = self._d2h(getattr(cols, name), n, _MEMOP_DTYPE[name])
        return out

    def memory_witness(self) -> MemoryWitnessC:
        """zkir_result_memory_trace: device columns (row order, CSR row offsets, sorted)."""
        w = MemoryWitnessC()
        rc = lib().zkir_result_memory_trace(self._r, C.byref(w))
        if rc != ZKIR_OK:
            _raise(rc)
        return w

    def get_memory_trace(self) -> np.ndarray:
        """ExecutionResult::get_memory_trace (vm.rs:85-94): every data-memory op, stably sorted by (timestamp, address, Read<Write).
        Returns packed MemoryOp records (address, value, timestamp, is_write, width, bound_*)."""
        if not self._r or self._log.n_rows == 0:
            return np.zeros(0, dtype=_MEMOP_DTYPE)
        w = self.memory_witness()
        return self._memops(w.sorted, w.n_ops)

    def row_memory_ops(self) -> Tuple[np.ndarray, np.ndarray]:
        """TraceRow.memory_ops of every row (trace.rs:49): (ops in row order, offsets[n_rows+1])."""
        if not self._r or self._log.n_rows == 0:
            return np.zeros(0, dtype=_MEMOP_DTYPE), np.zeros(1, dtype=np.uint64)
        w = self.memory_witness()
        return self._memops(w.row_order, w.n_ops), self._d2h(w.row_offsets, w.n_rows + 1, "<u8")

    def memory_op_count(self) -> int:
        """ExecutionResult::memory_op_count (vm.rs:97-102)."""
        return len(self._log.mem_events)

    def range_check_witness(self) -> Optional[RangeCheckWitnessC]:
        if not self._r:
            return None
        w = RangeCheckWitnessC()
        rc = lib().zkir_result_range_check_witnesses(self._r, C.byref(w))
        if rc != ZKIR_OK:
            _raise(rc)
        return w

    @property
    def range_check_witnesses(self) -> list:
        """Vec<RangeCheckWitness> (range_check.rs:209-238): one list of (value, chunks[4], pc) per non-empty checkpoint."""
        log = self._log
        if len(log.rc_events) == 0:
            return []
        if self._r:
            w = self.range_check_witness()
            n = w.n_checks
            v, p = self._d2h(w.value, n, "<u8"), self._d2h(w.pc, n, "<u8")
            c = self._d2h(w.chunks, 4 * w.chunk_stride, "<u2").reshape(4, -1)[:, :n].T
        else:                                   # run without an execution trace: no device handle; expand through the layer-2 launch
            from . import pipeline as pl
            value, pc, chunks, _ = pl.range_checks(log)
            v, p, c = value.cpu().numpy().view(np.uint64), pc.cpu().numpy().view(np.uint64), chunks.cpu().numpy().view(np.uint16).T
        offs = log.rc_offsets
        return [[(int(v[i]), [int(x) for x in c[i]], int(p[i])) for i in range(int(offs[k]), int(offs[k + 1]))] for k in range(len(offs) - 1)]

    @property
    def normalization_witnesses(self) -> np.ndarray:
        """Vec<NormalizationEvent> (normalization_witness.rs:129-138) as packed records."""
        if len(self._log.norm_events) == 0:
            return np.zeros(0, dtype=_NORM_DTYPE)
        if not self._r:
            from . import pipeline as pl
            return pl.normalization_events(self._log)
        w = NormalizationWitnessC()
        rc = lib().zkir_result_normalization_witnesses(self._r, C.byref(w))
        if rc != ZKIR_OK:
            _raise(rc)
        n, c = w.n_events, w.columns
        out = np.zeros(n, dtype=_NORM_DTYPE)
        out["cycle"] = self._d2h(c.cycle, n, "<u8"); out["pc"] = self._d2h(c.pc, n, "<u8")
        out["reg"] = self._d2h(c.reg, n, "u1"); out["opcode"] = self._d2h(c.opcode, n, "u1")
        out["accumulated"][:, 0] = self._d2h(c.accumulated0, n, "<u8"); out["accumulated"][:, 1] = self._d2h(c.accumulated1, n, "<u8")
        out["normalized"][:, 0] = self._d2h(c.normalized0, n, "<u4"); out["normalized"][:, 1] = self._d2h(c.normalized1, n, "<u4")
        out["carries"][:, 0] = self._d2h(c.carry0, n, "<u4"); out["carries"][:, 1] = self._d2h(c.carry1, n, "<u4")
        out["normalized_bits"] = 20; out["limb_bits"] = 30; out["cause"] = 0
        return out

    def sha256_witnesses(self) -> Tuple[np.ndarray, np.ndarray]:
        """Sha256Witness columns of every single-block SHA-256 syscall: (uint32[608][n_blocks], timestamps[n_blocks])."""
        if not self._r:
            return np.zeros((608, 0), dtype=np.uint32), np.zeros(0, dtype=np.uint64)
        w = Sha256WitnessC()
        rc = lib().zkir_result_sha256_witnesses(self._r, C.byref(w))
        if rc != ZKIR_OK:
            _raise(rc)
        cols = self._d2h(w.columns, 608 * w.stride, "<u4").reshape(608, -1)[:, :w.n_blocks]
        return cols, self._d2h(w.timestamps, w.n_blocks, "<u8")

    def exec_stage_ms(self) -> dict:
        """Wall-clock breakdown of the zkir_exec call behind this result (zkir_result_stage_ms)."""
        ms = (C.c_float * 4)()
        if self._r:
            lib().zkir_result_stage_ms(self._r, ms)
        return dict(zip(("interpret", "device_alloc", "h2d", "k1_and_sync"), [float(x) for x in ms]))

    def public_inputs(self) -> PublicInputsC:
        """Public inputs of a proof of this run (zkir_public_inputs_of)."""
        return public_inputs(self._log, self._blob, self._inputs, self._config.enable_deferred_model)

    def close(self):
        if self._r:
            lib().zkir_result_free(self._r)    # also frees the delta log it owns
            self._r = None
            self._log._h = None
        else:
            self._log.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VM:
    """VM::new + VM::run (vm.rs:138-358).  `run()` consumes the VM, as in the reference."""

    def __init__(self, program: Program | bytes, inputs: Sequence[int] = (), config: Optional[VMConfig] = None):
        self._blob = bytes(program) if isinstance(program, (bytes, bytearray)) else program.to_bytes()
        self._inputs = [int(x) & (2**64 - 1) for x in inputs]
        self._config = config or VMConfig()
        self._consumed = False

    def run_window(self, row_begin: int, row_end: int) -> ExecutionResult:
        """zkir_exec_window: one GPU's share of the run — rows before row_begin executed untraced on this thread, rows [row_begin,
        row_end) traced, uploaded and filled on the current device (multi-GPU: every rank calls it with its own range)."""
        if self._consumed:
            raise ValueError("VM::run consumes the VM (vm.rs:208)")
        self._consumed = True
        L = lib()
        cfg = self._config._c()
        arr = (C.c_uint64 * max(1, len(self._inputs)))(*self._inputs)
        out = C.c_void_p()
        rc = L.zkir_exec_window(self._blob, len(self._blob), arr, len(self._inputs), C.byref(cfg), int(row_begin), int(row_end), C.byref(out))
        if rc != ZKIR_OK:
            _raise(rc)
        log = DeltaLog(L.zkir_result_delta_log(out.value), owned=False)
        return ExecutionResult(out.value, log, self._blob, self._inputs, self._config)

    def run(self) -> ExecutionResult:
        if self._consumed:
            raise ValueError("VM::run consumes the VM (vm.rs:208)")
        self._consumed = True
        L = lib()
        cfg = self._config._c()
        arr = (C.c_uint64 * max(1, len(self._inputs)))(*self._inputs)
        if not self._config.enable_execution_trace:
            # nothing to materialise on the device: host stage only
            return ExecutionResult(None, interpret(self._blob, self._inputs, self._config), self._blob, self._inputs, self._config)
        out = C.c_void_p()
        rc = L.zkir_exec(self._blob, len(self._blob), arr, len(self._inputs), C.byref(cfg), C.byref(out))
        if rc != ZKIR_OK:
            _raise(rc)
        log = DeltaLog(L.zkir_result_delta_log(out.value), owned=False)
        return ExecutionResult(out.value, log, self._blob, self._inputs, self._config)


def exec_shard(log: "DeltaLog", row_begin: int, row_end: int, blob: bytes = b"", inputs: Sequence[int] = (), config: Optional[VMConfig] = None) -> ExecutionResult:
    """zkir_exec_shard: the drop-in handle for rows [row_begin, row_end) of a finished interpretation, cut out, uploaded to the current
    device and filled there (multi-GPU: one call per device; segment proofs: ranges that share one row)."""
    L = lib()
    out = C.c_void_p()
    rc = L.zkir_exec_shard(log._h, int(row_begin), int(row_end), C.byref(out))
    if rc != ZKIR_OK:
        _raise(rc)
    return ExecutionResult(out.value, DeltaLog(L.zkir_result_delta_log(out.value), owned=False), blob, inputs, config)


def run(program: Program | bytes, inputs: Sequence[int] = ()) -> List[int]:
    """zkir_runtime::run (lib.rs:59-62): default config, outputs only."""
    res = VM(program, inputs, VMConfig()).run()
    try:
        return list(res.outputs)
    finally:
        res.close()
