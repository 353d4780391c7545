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
    """Upload raw bytes of a (possibly structured) numpy array as a uint8 tensor."""
    flat = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    if flat.size == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.from_numpy(flat.copy()).to(device, non_blocking=False)


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
    pc = torch.zeros(cap, dtype=torch.int64, device=device)
    inst = torch.zeros(cap, dtype=torch.int32, device=device)
    if log.n_rows:
        pc[:log.n_rows] = torch.from_numpy(log.pc.view(np.int64).copy()).to(device)
        inst[:log.n_rows] = torch.from_numpy(log.inst.view(np.int32).copy()).to(device)
    return DeviceDeltaLog(n_rows=log.n_rows, cycle_base=log.cycle_base, tile_rows=T, n_tiles=log.n_tiles, n_events=len(log.reg_events),
                          events=_to_dev(log.reg_events, device), tile_ev_off=_to_dev(log.tile_ev_off, device),
                          tile_snap=_to_dev(log.tile_snap, device), pc=pc, inst=inst)


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
