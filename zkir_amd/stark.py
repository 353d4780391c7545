"""Driver for the self-defined prover stages (stark.hip) — Baby Bear LDE + Poseidon2-12 Merkle commitment.

NOT in the reference (parity unpinned); spec = oracle/stark_oracle.cpp, DESIGN.md §8.
PyTorch is plumbing (device buffers, streams); all arithmetic happens in the HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import pipeline as pl
from . import runtime as rt

P = 2013265921
W_MAIN = 152                 # COMMITTED main-trace columns of a default-mode run (zkir_main_trace_width); the AIR has 172 logical columns (air.h: W)
W_MAIN_DEFERRED = 168        # deferred mode: the storage states are committed too
W_AUX = 40                  # aux trace of the lookup argument (air.h): H0..H7, HR, S as four base columns each
RC_TABLE = 1024
HEADER_WORDS = 157


MEM_MULT = RC_TABLE + 256 + 16 + 3 * 256 + RC_TABLE     # (mode 3) LOW3 | BYTE | NIBBLE | AND | OR | XOR | LOW6 multiplicities (air.h MEM_MULT)


def proof_layout(proof) -> dict:
    """Word offsets inside a proof of ANY mode (word 9): header (157 words; modes 2 / 3: + the four counter words) | program (byte length, 16-bit halfwords) | modes 2 / 3: the
    I/O section (n_in, inputs as four 16-bit pieces each, n_out, outputs, halt kind, halt code) | mode 3: the touched cells (n, 7 words a cell) | mode 4: the hash tape, the wide tape | ROM multiplicities | range
    multiplicities | mode 3: the LOW3 .. LOW6 multiplicities | trace root | aux root | quotient root | openings ...  (oracle/stark_oracle.cpp so::prove is the layout's definition)."""
    mode = int(proof[9])
    at = HEADER_WORDS + (4 if mode >= 2 else 0)
    blob_len = int(proof[at])
    at += 1
    half = np.asarray(proof[at:at + (blob_len + 1) // 2], dtype=np.uint32)
    blob = np.stack([half & 0xFF, half >> 8], axis=1).astype(np.uint8).reshape(-1)[:blob_len].tobytes()
    n_rom = int.from_bytes(blob[16:20], "little") // 4 if blob_len >= 32 else 0
    at += (blob_len + 1) // 2
    io_at = mem_at = None
    if mode >= 2:
        io_at = at
        n_in = int(proof[at]); at += 1 + 4 * n_in
        n_out = int(proof[at]); at += 1 + 4 * n_out
        at += 5
    hash_at = wide_at = None
    if mode >= 3:
        mem_at = at
        at += 1 + 7 * int(proof[at])
    if mode == 4:                                             # (mode 4) the hash calls: [n] then per call 8 words + 5 per touched cell
        hash_at = at
        n_calls = int(proof[at]); at += 1
        for _ in range(n_calls):
            at += 8 + 5 * int(proof[at + 7])
        wide_at = at                                          # .. then the wide tape: [n] then per record [cycle] [rs1: three limbs] [rs2: three limbs] [opcode]
        at += 1 + 8 * int(proof[at])
    rom_mult = at
    troot = rom_mult + n_rom + RC_TABLE + (MEM_MULT if mode >= 3 else 0)
    return {"mode": mode, "num_queries": int(proof[4]), "pow_bits": int(proof[6]), "blob": blob, "n_rom": n_rom, "io_section": io_at, "mem_section": mem_at, "hash_section": hash_at, "wide_section": wide_at, "rom_mult": rom_mult,
            "rc_mult": rom_mult + n_rom, "trace_root": troot, "aux_root": troot + 4, "quotient_root": troot + 8, "openings": troot + 12}


def trace_root(proof) -> list:
    t0 = proof_layout(proof)["trace_root"]
    return [int(x) for x in proof[t0:t0 + 4]]


class StarkContext:
    """zkir_stark_ctx: device tables for traces of 2^log_n rows (blow-up 2)."""

    def __init__(self, log_n: int, log_blowup: int = 1):
        pl._require_gpu()
        self.log_n, self.log_blowup = log_n, log_blowup
        h = C.c_void_p()
        rc = rt.lib().zkir_stark_ctx_create(log_n, log_blowup, C.byref(h))
        if rc != rt.ZKIR_OK:
            rt._raise(rc)
        self._h = h

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            rt.lib().zkir_stark_ctx_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _sp(stream):
    return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)


def padded_log_n(n_real: int) -> int:
    return int(rt.lib().zkir_padded_log_n(n_real))


# ---- the B8 matrix layout of the C ABI (include/zkir_amd.h): blocks of 8 columns, block b = [rows][8] -----------------------------
def to_b8(cols: torch.Tensor) -> torch.Tensor:
    """Column-major int32[width][n] -> B8 int32[ceil(width/8)][n][8] (missing columns of the last block are zero)."""
    width, n = cols.shape
    nb = (width + 7) // 8
    out = torch.zeros((nb * 8, n), dtype=cols.dtype, device=cols.device)
    out[:width] = cols
    return out.view(nb, 8, n).permute(0, 2, 1).contiguous()


def main_width(deferred: bool = False) -> int:
    """Committed main-trace columns of a run (zkir_main_trace_width_for): 152 in the default VM mode, 168 with the deferred model."""
    return W_MAIN_DEFERRED if deferred else W_MAIN


def from_b8(mat: torch.Tensor, width: int) -> torch.Tensor:
    """B8 int32[nb][n][8] -> column-major int32[width][n]."""
    nb, n, _ = mat.shape
    return mat.permute(0, 2, 1).reshape(nb * 8, n)[:width].contiguous()


def main_trace(trace: pl.DeviceTrace, stream=None, deferred: bool = False) -> torch.Tensor:
    """K4: SoA execution trace (n_rows executed rows) -> the COMMITTED main-trace matrix in the B8 layout, int32[main_width(deferred) / 8][N][8],
    N = the padded power of two."""
    n = trace.n_rows
    out = torch.empty((main_width(deferred) // 8, 1 << padded_log_n(n), 8), dtype=torch.int32, device=trace.cycle.device)
    pl._check(rt.lib().zkir_main_trace_launch(C.byref(trace.c), n, int(deferred), out.data_ptr(), _sp(stream)))
    return out


def lde(ctx: StarkContext, mat: torch.Tensor, stream=None, clobber: bool = False) -> torch.Tensor:
    """Per-column LDE of a B8 matrix int32[nb][N][8] to int32[nb][2N][8] on the coset 31*<w_2N> (natural order)."""
    nb, n, eight = mat.shape
    assert eight == 8 and n == 1 << ctx.log_n and mat.dtype == torch.int32 and mat.is_contiguous()
    src = mat if clobber else mat.clone()
    out = torch.empty((nb, 2 * n, 8), dtype=torch.int32, device=mat.device)
    pl._check(rt.lib().zkir_lde_launch(ctx.handle, src.data_ptr(), nb * 8, out.data_ptr(), _sp(stream)))
    return out


def merkle_commit(ctx: StarkContext, mat: torch.Tensor, width: Optional[int] = None, stream=None) -> torch.Tensor:
    """Poseidon2-12 Merkle tree over the rows (positions) of a B8 matrix int32[nb][n][8] with `width` real columns (default all);
    returns the tree int32[4*(2n-1)], root = last 4."""
    nb, n, eight = mat.shape
    assert eight == 8 and mat.dtype == torch.int32 and mat.is_contiguous()
    width = nb * 8 if width is None else width
    tree = torch.empty(4 * (2 * n - 1), dtype=torch.int32, device=mat.device)
    pl._check(rt.lib().zkir_merkle_commit_launch(ctx.handle, mat.data_ptr(), width, n, tree.data_ptr(), _sp(stream)))
    return tree


def merkle_cap(ctx: StarkContext, digests: torch.Tensor, stream=None) -> torch.Tensor:
    """Root over n (power of two) digests int32[n][4] — the all-gathered per-GPU subtree roots of a row-sharded commitment."""
    n = digests.shape[0]
    tree = torch.empty(4 * (2 * n - 1), dtype=torch.int32, device=digests.device)
    tree[:4 * n] = digests.reshape(-1)
    pl._check(rt.lib().zkir_merkle_cap_launch(ctx.handle, tree.data_ptr(), n, _sp(stream)))
    return tree[-4:]


def commit_trace(ctx: StarkContext, trace: pl.DeviceTrace, stream=None, deferred: bool = False):
    """main trace -> LDE -> Merkle.  Returns (root np.uint32[4], lde matrix tensor (B8), tree tensor)."""
    m = main_trace(trace, stream, deferred)
    L = lde(ctx, m, stream, clobber=True)
    tree = merkle_commit(ctx, L, main_width(deferred), stream)
    root = tree[-4:].cpu().numpy().view(np.uint32)
    return root, L, tree


def prove(ctx: StarkContext, trace, pub: rt.PublicInputsC, stream=None, want_stage_ms: bool = False):
    """zkir_prove: full ZKIR-STARK v1 proof of a device trace (`trace`: pl.DeviceTrace or zkir_trace_columns) bound to the public
    inputs `pub` (rt.public_inputs / ExecutionResult.public_inputs).  Returns np.uint32 proof words (and stage ms)."""
    cols = trace.c if hasattr(trace, "c") else trace
    out = C.POINTER(C.c_uint32)()
    n_words = C.c_uint64()
    ms = (C.c_float * 9)()
    pl._check(rt.lib().zkir_prove(ctx.handle, C.byref(cols), C.byref(pub), C.byref(out), C.byref(n_words), ms if want_stage_ms else None, _sp(stream)))
    proof = np.ctypeslib.as_array(out, shape=(n_words.value,)).copy()
    rt.lib().zkir_proof_free(out)
    return (proof, list(ms)) if want_stage_ms else proof


verify = rt.verify
