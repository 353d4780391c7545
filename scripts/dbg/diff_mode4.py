import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import test_gpu_stark as T
from oracle import stark_api as so
from zkir_amd import stark, runtime as rt
for which in sys.argv[1:] or ["wide_grid"]:
    blob, ins, ores, log, tr, opub, pub = T._mode3_case(which, "device", wide=True)
    ctx = stark.StarkContext(stark.padded_log_n(len(ores.rows)))
    proof = stark.prove(ctx, tr, pub)
    want = so.prove(ores.rows, opub)
    lay = stark.proof_layout(want)
    bad = np.nonzero(proof[:min(len(proof), len(want))] != want[:min(len(proof), len(want))])[0]
    print(which, len(proof), len(want), "first diff", bad[:5], {k: v for k, v in lay.items() if k != "blob"})
    print("verify", rt.verify(proof), so.verify(proof))
    for b in bad[:8]: print("  word", int(b), "gpu", int(proof[b]), "oracle", int(want[b]))
    M = so.main_trace(ores.rows, opub)
    host = rt.main_trace_wide_host(tr, len(ores.rows), pub) if hasattr(rt, "main_trace_wide_host") else None
    got_rows = tr.rows()
    for name in ores.rows.dtype.names:
        if not np.array_equal(got_rows[name], ores.rows[name]):
            print("  TRACE differs in", name)
    import ctypes as C, torch
    N = 1 << so.padded_log_n(len(ores.rows)); wm = so.committed_width(4)
    out = torch.zeros((wm // 8, N, 8), dtype=torch.int32, device="cuda")
    scratch = torch.zeros(2 * N + 2 * (N // 1024 + 2) + 64, dtype=torch.int32, device="cuda")
    tape = torch.tensor(list(ins) if len(ins) else [0], dtype=torch.int64, device="cuda")
    class IoArgs(C.Structure):
        _fields_ = [("inputs", C.c_void_p), ("n_inputs", C.c_uint64), ("writes_before", C.c_uint64), ("reads_before", C.c_uint64)]
    io = IoArgs(tape.data_ptr(), len(ins), 0, 0)
    hpub = rt.public_inputs(log, blob, list(ins), wide_mode=True, mem_witness="host")
    nr = len(ores.rows)
    mo = torch.from_numpy(np.ctypeslib.as_array(C.cast(hpub.mem_old, C.POINTER(C.c_uint64)), (nr,)).astype(np.int64)).cuda()
    mt = torch.from_numpy(np.ctypeslib.as_array(C.cast(hpub.mem_told, C.POINTER(C.c_uint32)), (nr,)).astype(np.int32)).cuda()
    L = rt.lib()
    L.zkir_main_trace_wide_launch.restype = C.c_int
    L.zkir_main_trace_wide_launch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    cols = tr.c
    rc = L.zkir_main_trace_wide_launch(C.byref(cols), nr, C.byref(io), mo.data_ptr(), mt.data_ptr(), int.from_bytes(blob[16:20], "little"), scratch.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32).transpose(0, 2, 1).reshape(wm, N)
    want_m = so.to_committed(so.main_trace(ores.rows, opub), 4)
    nbad = 0
    for k in range(wm):
        if not np.array_equal(got[k], want_m[k]):
            r = int(np.nonzero(got[k] != want_m[k])[0][0]); nbad += 1
            if nbad < 8: print("  rc", rc, "committed column", k, "row", r, "gpu", int(got[k][r]), "oracle", int(want_m[k][r]))
    print("  columns differing:", nbad)
