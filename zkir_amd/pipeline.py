"""Layer-2 driver: device-resident delta log + explicit kernel launches on a chosen HIP stream.

PyTorch is plumbing here (device allocations, streams, events, torch.distributed); every computation
is a hand-written HIP kernel reached through the C ABI (include/zkir_amd.h).  Used by bench.py, by the
multi-GPU row-sharded path and by the GPU parity tests.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import runtime as rt


def _require_gpu():
    if not torch.cuda.is_available():
        raise rt.RuntimeError(rt.ERR_DEVICE, "zkir_amd.pipeline needs a HIP device (no CPU fallback)")


def _to_dev(a: np.ndarray, device) -> torch.Tensor:
    """Upload raw bytes of a (possibly structured) numpy array as a uint8 tensor: one hipMemcpy straight from the delta log's host
    buffer (zkir_host_to_device, the copy path of zkir_exec), ordered on torch's current stream."""
    flat = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    out = torch.empty(flat.size, dtype=torch.uint8, device=device)
    if flat.size:
        _check(rt.lib().zkir_host_to_device(out.data_ptr(), flat.ctypes.data, flat.size, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


@dataclass
class DeviceDeltaLog:
    """The delta log resident in HBM (what the timed region of bench.py starts from)."""
    n_rows: int
    cycle_base: int
    tile_rows: int
    n_tiles: int
    n_events: int
    events: torch.Tensor
    tile_ev_off: torch.Tensor
    tile_snap: torch.Tensor
    pc: torch.Tensor           # already in final column form
    inst: torch.Tensor


def upload(log: rt.DeltaLog, device=None) -> DeviceDeltaLog:
    _require_gpu()
    device = device or torch.device("cuda", torch.cuda.current_device())
    T = log.tile_rows
    cap = (log.n_rows + T - 1) // T * T
    pc = torch.zeros(cap, dtype=torch.int64, device=device) if cap != log.n_rows else torch.empty(cap, dtype=torch.int64, device=device)
    inst = torch.zeros(cap, dtype=torch.int32, device=device) if cap != log.n_rows else torch.empty(cap, dtype=torch.int32, device=device)
    if log.n_rows:
        pc[:log.n_rows].view(torch.uint8).copy_(_to_dev(log.pc, device))
        inst[:log.n_rows].view(torch.uint8).copy_(_to_dev(log.inst, device))
    ddl = DeviceDeltaLog(n_rows=log.n_rows, cycle_base=log.cycle_base, tile_rows=T, n_tiles=log.n_tiles, n_events=len(log.reg_events),
                         events=_to_dev(log.reg_events, device), tile_ev_off=_to_dev(log.tile_ev_off, device),
                         tile_snap=_to_dev(log.tile_snap, device), pc=pc, inst=inst)
    # The log's big buffers are PINNED pool blocks (csrc/host.h), so these copies are asynchronous DMA: the log must not give its blocks back to the pool before the
    # stream has passed them.  DeltaLog.close() waits for the events recorded here.
    if hasattr(log, "_uploads"):
        ev = torch.cuda.Event()
        ev.record()
        log._uploads.append(ev)
    return ddl


class HostShard:
    """The part of a (sharded) delta log K1 needs, as plain numpy arrays: what `save_shard` / `load_shard` move between the
    processes of one node (bench.py --gpus N: rank 0 interprets once, every rank picks up its row range)."""
    FIELDS = ("pc", "inst", "reg_events", "tile_ev_off", "tile_snap")

    def __init__(self, n_rows, cycle_base, tile_rows, n_tiles, **arrays):
        self.n_rows, self.cycle_base, self.tile_rows, self.n_tiles = int(n_rows), int(cycle_base), int(tile_rows), int(n_tiles)
        for k in self.FIELDS:
            setattr(self, k, arrays[k])


def save_shard(shard, path: str) -> None:
    """Write a delta log (usually `log.shard(a, b)`) to `path` (.npz, uncompressed; /dev/shm keeps it in memory)."""
    np.savez(path, meta=np.array([shard.n_rows, shard.cycle_base, shard.tile_rows, shard.n_tiles], dtype=np.uint64),
             **{k: np.ascontiguousarray(getattr(shard, k)) for k in HostShard.FIELDS})


def load_shard(path: str) -> HostShard:
    with np.load(path) as z:
        meta = z["meta"]
        return HostShard(meta[0], meta[1], meta[2], meta[3], pc=z["pc"], inst=z["inst"], reg_events=z["reg_events"].view(rt.REG_EVENT_DTYPE).reshape(-1),
                         tile_ev_off=z["tile_ev_off"], tile_snap=z["tile_snap"].reshape(-1, 16))


class DeviceTrace:
    """Wide SoA execution trace in HBM (zkir_trace_columns); columns are torch tensors sharing one allocation pattern."""

    def __init__(self, ddl: DeviceDeltaLog):
        _require_gpu()
        dev = ddl.events.device
        T = ddl.tile_rows
        self.n_rows = ddl.n_rows
        self.cap = cap = (ddl.n_rows + T - 1) // T * T
        self.cycle = torch.empty(cap, dtype=torch.int64, device=dev)
        self.pc, self.instruction = ddl.pc, ddl.inst                     # zero-copy: delivered in final form
        self.registers = torch.empty((16, cap), dtype=torch.int64, device=dev)
        self.bound_bits = torch.empty((16, cap), dtype=torch.int32, device=dev)
        self.bound_tag = torch.empty((16, cap), dtype=torch.uint8, device=dev)
        self.bound_payload = torch.empty((16, cap), dtype=torch.int64, device=dev)
        self.reg_state = torch.empty((16, cap), dtype=torch.uint8, device=dev)
        self.c = rt.TraceColumnsC(self.cycle.data_ptr(), self.pc.data_ptr(), self.instruction.data_ptr(), self.registers.data_ptr(),
                                  self.bound_bits.data_ptr(), self.bound_tag.data_ptr(), self.bound_payload.data_ptr(),
                                  self.reg_state.data_ptr(), cap)

    def rows(self) -> np.ndarray:
        """Copy to host as reference-shaped packed rows (tests)."""
        from .runtime import ExecutionTrace  # noqa: F401  (dtype lives there)
        n = self.n_rows
        dt = np.dtype([("cycle", "<u8"), ("pc", "<u8"), ("instruction", "<u4"), ("registers", "<u8", (16,)),
                       ("bound_bits", "<u4", (16,)), ("bound_tag", "u1", (16,)), ("bound_payload", "<u8", (16,)), ("reg_state", "u1", (16,))])
        out = np.zeros(n, dtype=dt)
        out["cycle"] = self.cycle[:n].cpu().numpy().view(np.uint64)
        out["pc"] = self.pc[:n].cpu().numpy().view(np.uint64)
        out["instruction"] = self.instruction[:n].cpu().numpy().view(np.uint32)
        out["registers"] = self.registers[:, :n].cpu().numpy().view(np.uint64).T
        out["bound_bits"] = self.bound_bits[:, :n].cpu().numpy().view(np.uint32).T
        out["bound_tag"] = self.bound_tag[:, :n].cpu().numpy().T
        out["bound_payload"] = self.bound_payload[:, :n].cpu().numpy().view(np.uint64).T
        out["reg_state"] = self.reg_state[:, :n].cpu().numpy().T
        return out


def trace_fill_args(ddl: DeviceDeltaLog, trace: DeviceTrace) -> rt.TraceFillArgsC:
    return rt.TraceFillArgsC(ddl.events.data_ptr(), ddl.tile_ev_off.data_ptr(), ddl.tile_snap.data_ptr(), ddl.n_rows, ddl.cycle_base,
                             ddl.tile_rows, ddl.n_events, trace.c)


def trace_fill(args: rt.TraceFillArgsC, stream: Optional[torch.cuda.Stream] = None) -> None:
    """K1 launch (asynchronous) on `stream` (default: torch's current stream, so torch.cuda.Event brackets it)."""
    s = stream or torch.cuda.current_stream()
    rc = rt.lib().zkir_trace_fill_launch(C.byref(args), C.c_void_p(s.cuda_stream))
    if rc != rt.ZKIR_OK:
        rt._raise(rc)


def trace_fill_bytes(ddl: DeviceDeltaLog) -> int:
    return int(rt.lib().zkir_trace_fill_bytes(ddl.n_rows, ddl.n_events, ddl.n_tiles))


# ---- witness expansion (witness.hip) -----------------------------------------------------------------
def _stream_ptr(stream):
    return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)


def _check(rc):
    if rc != rt.ZKIR_OK:
        rt._raise(rc)


class MemopColumns:
    """zkir_memop_columns backed by torch tensors."""
    FIELDS = (("address", torch.int64), ("value", torch.int64), ("timestamp", torch.int64), ("is_write", torch.uint8), ("width", torch.uint8),
              ("bound_bits", torch.int32), ("bound_tag", torch.uint8), ("bound_payload", torch.int64))

    def __init__(self, n: int, device):
        self.n = n
        self.t = {name: torch.empty(max(n, 1), dtype=dt, device=device) for name, dt in self.FIELDS}
        self.c = rt.MemopColumnsC(*[self.t[name].data_ptr() for name, _ in self.FIELDS])

    def to_numpy(self) -> np.ndarray:
        """Packed reference-shaped MemoryOp records (same dtype as the oracle's)."""
        dt = np.dtype([("address", "<u8"), ("value", "<u8"), ("timestamp", "<u8"), ("is_write", "u1"), ("width", "u1"),
                       ("bound_bits", "<u4"), ("bound_tag", "u1"), ("bound_payload", "<u8")])
        out = np.zeros(self.n, dtype=dt)
        for name, _ in self.FIELDS:
            a = self.t[name][:self.n].cpu().numpy()
            out[name] = a.view(dt[name]) if a.dtype.itemsize == dt[name].itemsize else a
        return out


def memory_ops(log: rt.DeltaLog, device=None, stream=None, separate_passes: bool = False):
    """Returns (row_order MemopColumns, row_offsets tensor[n_rows+1], sorted MemopColumns = get_memory_trace()).
    Uploads, allocations and launches all happen on `stream` (default: the current stream), which is synchronised before returning.
    Default: two passes over the events (zkir_memops_expand_csr_launch: expansion + CSR offsets + shape flags; then the sort);
    separate_passes=True drives the stand-alone entry points instead (binary-search CSR, expansion, sort with its own check pass)."""
    _require_gpu()
    device = device or torch.device("cuda", torch.cuda.current_device())
    L = rt.lib()
    n, n_rows = len(log.mem_events), log.n_rows
    s = stream or torch.cuda.current_stream()
    with torch.cuda.stream(s):
        ev = _to_dev(log.mem_events, device)
        sp = _stream_ptr(s)
        row_cols, sorted_cols = MemopColumns(n, device), MemopColumns(n, device)
        offsets = torch.empty(n_rows + 1, dtype=torch.int64, device=device)
        scratch = torch.empty(max(n_rows, 1), dtype=torch.uint8, device=device)
        if separate_passes:
            _check(L.zkir_memops_row_offsets_launch(ev.data_ptr(), n, n_rows, offsets.data_ptr(), sp))
            _check(L.zkir_memops_expand_launch(ev.data_ptr(), n, log.cycle_base, C.byref(row_cols.c), sp))
            _check(L.zkir_memops_sort_launch(ev.data_ptr(), n, n_rows, log.cycle_base, offsets.data_ptr(), scratch.data_ptr(), C.byref(sorted_cols.c), sp))
        else:
            _check(L.zkir_memops_expand_csr_launch(ev.data_ptr(), n, n_rows, log.cycle_base, C.byref(row_cols.c), offsets.data_ptr(), scratch.data_ptr(), sp))
            _check(L.zkir_memops_sort_prepared_launch(ev.data_ptr(), n, log.cycle_base, offsets.data_ptr(), scratch.data_ptr(), C.byref(sorted_cols.c), sp))
        s.synchronize()
    return row_cols, offsets, sorted_cols


def range_checks(log: rt.DeltaLog, device=None, stream=None):
    """Returns (value[n], pc[n], chunks[4][n] u16 as int16 tensor, multiplicity[2^chunk_bits])."""
    _require_gpu()
    device = device or torch.device("cuda", torch.cuda.current_device())
    n = len(log.rc_events)
    s = stream or torch.cuda.current_stream()
    with torch.cuda.stream(s):
        ev = _to_dev(log.rc_events, device)
        value = torch.empty(max(n, 1), dtype=torch.int64, device=device)
        pc = torch.empty(max(n, 1), dtype=torch.int64, device=device)
        chunks = torch.empty((4, max(n, 1)), dtype=torch.int16, device=device)
        mult = torch.empty(1 << log.rc_chunk_bits, dtype=torch.int32, device=device)
        _check(rt.lib().zkir_range_check_expand_launch(ev.data_ptr(), n, log.rc_chunk_bits, value.data_ptr(), pc.data_ptr(), chunks.data_ptr(),
                                                        max(n, 1), mult.data_ptr(), _stream_ptr(s)))
        s.synchronize()
    return value[:n], pc[:n], chunks[:, :n], mult


def normalization_events(log: rt.DeltaLog, device=None, stream=None) -> np.ndarray:
    """NormalizationEvent records (oracle NORM_DTYPE layout) computed on the device."""
    _require_gpu()
    device = device or torch.device("cuda", torch.cuda.current_device())
    n = len(log.norm_events)
    s = stream or torch.cuda.current_stream()
    spec_ = (("cycle", torch.int64), ("pc", torch.int64), ("reg", torch.uint8), ("opcode", torch.uint8), ("accumulated0", torch.int64),
             ("accumulated1", torch.int64), ("normalized0", torch.int32), ("normalized1", torch.int32), ("carry0", torch.int32), ("carry1", torch.int32))
    with torch.cuda.stream(s):
        ev = _to_dev(log.norm_events, device)
        t = {k: torch.empty(max(n, 1), dtype=dt, device=device) for k, dt in spec_}
        cols = rt.NormColumnsC(*[t[k].data_ptr() for k, _ in spec_])
        _check(rt.lib().zkir_norm_expand_launch(ev.data_ptr(), n, C.byref(cols), _stream_ptr(s)))
        s.synchronize()
    dt = np.dtype([("cycle", "<u8"), ("pc", "<u8"), ("reg", "u1"), ("accumulated", "<u8", (2,)), ("normalized", "<u4", (2,)),
                   ("carries", "<u4", (2,)), ("normalized_bits", "u1"), ("limb_bits", "u1"), ("cause", "u1"), ("opcode", "u1")])
    out = np.zeros(n, dtype=dt)
    g = lambda k: t[k][:n].cpu().numpy()  # noqa: E731
    out["cycle"] = g("cycle").view(np.uint64); out["pc"] = g("pc").view(np.uint64); out["reg"] = g("reg"); out["opcode"] = g("opcode")
    out["accumulated"][:, 0] = g("accumulated0").view(np.uint64); out["accumulated"][:, 1] = g("accumulated1").view(np.uint64)
    out["normalized"][:, 0] = g("normalized0").view(np.uint32); out["normalized"][:, 1] = g("normalized1").view(np.uint32)
    out["carries"][:, 0] = g("carry0").view(np.uint32); out["carries"][:, 1] = g("carry1").view(np.uint32)
    out["normalized_bits"] = 20; out["limb_bits"] = 30; out["cause"] = 0
    return out


def sha256_chip(blocks: np.ndarray, device=None, stream=None):
    """K3: blocks = rt.SHA_BLOCK_DTYPE records -> (columns tensor int32 [608][n], timestamps int64[n]); `stream` is synchronised."""
    _require_gpu()
    device = device or torch.device("cuda", torch.cuda.current_device())
    n = len(blocks)
    s = stream or torch.cuda.current_stream()
    with torch.cuda.stream(s):
        ev = _to_dev(blocks, device)
        stride = (max(n, 1) + 63) // 64 * 64              # a multiple of 4 selects the 16-byte-store kernel
        out = torch.empty((608, stride), dtype=torch.int32, device=device)
        ts = torch.empty(max(n, 1), dtype=torch.int64, device=device)
        _check(rt.lib().zkir_sha256_chip_launch(ev.data_ptr(), n, out.data_ptr(), stride, ts.data_ptr(), _stream_ptr(s)))
        s.synchronize()
    return out[:, :n], ts[:n]
