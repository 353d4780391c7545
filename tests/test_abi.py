"""The C-ABI library loads and exports every symbol include/zkir_amd.h declares; host-only entry points
behave (error reporting, argument checks).  No compute on a device here."""
import ctypes as C
import os
import re

import pytest

from zkir_amd import runtime as rt, spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header="zkir_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkir_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    L = rt.lib()
    names = _declared_functions()
    assert len(names) > 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/zkir_amd.h but not exported: {missing}"
    # the probes / experiments live in their own header: exported too, and none of them is declared in the drop-in header
    exp = _declared_functions("zkir_amd_experimental.h")
    assert {"zkir_modmul_peak_per_s", "zkir_hbm_copy_peak_gbs", "zkir_commit_fused01_launch", "zkir_ntt_strided_variant_launch"} <= set(exp)
    assert not [n for n in exp if not hasattr(L, n)]
    assert not set(exp) & set(names)
    assert {"zkir_prove_result", "zkir_proof_bytes_free", "zkir_public_inputs_set_params", "zkir_proof_version_of_mode"} <= set(names)


def test_struct_sizes_match_header():
    assert C.sizeof(rt.VmConfigC) == 16
    assert rt.REG_EVENT_DTYPE.itemsize == 32 and rt.MEM_EVENT_DTYPE.itemsize == 24
    assert rt.RC_EVENT_DTYPE.itemsize == 16 and rt.NORM_EVENT_DTYPE.itemsize == 32 and rt.SHA_BLOCK_DTYPE.itemsize == 72
    assert C.sizeof(rt.TraceColumnsC) == 9 * 8


def test_version_and_last_error():
    L = rt.lib()
    assert b"zkir_amd" in L.zkir_version()
    with pytest.raises(rt.RuntimeError) as e:
        rt.interpret(b"\x00" * 10)
    assert e.value.code == rt.ERR_BAD_PROGRAM and "Invalid header size" in e.value.message


def test_interpret_argument_checks():
    blob = spec.fib_program(5).to_bytes()
    with pytest.raises(rt.RuntimeError) as e:
        rt.interpret(blob, config=rt.VMConfig(enable_execution_trace=True), tile_rows=300)
    assert e.value.code == rt.ERR_ARGUMENT


def test_reference_api_mirror_without_trace():
    """zkir_runtime::run / VM::new(..).run() with the default config never needs a device (vm.rs:40-50)."""
    assert rt.run(spec.fib_program(30)) == [832040]
    vm = rt.VM(spec.fib_program(5), [], rt.VMConfig())
    res = vm.run()
    assert res.outputs == [5] and res.halt_reason == rt.HaltReason.Exit(0) and len(res.execution_trace) == 0
    with pytest.raises(ValueError):
        vm.run()                       # run() consumes the VM
    res.close()


def test_exec_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rt.RuntimeError) as e:
        rt.VM(spec.fib_program(5), [], rt.VMConfig(enable_execution_trace=True)).run()
    assert e.value.code == rt.ERR_DEVICE


def test_host_poseidon2_matches_oracle():
    """zkir_poseidon2_permute (the product's host build of poseidon2.h: wide linear layers, lazy Montgomery products, constants
    folded into multiply-adds) against the oracle's textbook permutation, on random states and on the edges of the range."""
    import numpy as np
    from oracle import stark_api as so
    P = so.P
    rng = np.random.default_rng(12)
    states = [rng.integers(0, P, 12).astype(np.uint32) for _ in range(300)]
    states += [np.full(12, P - 1, np.uint32), np.zeros(12, np.uint32), np.ones(12, np.uint32), np.arange(12, dtype=np.uint32),
               np.array([P - 1, 0] * 6, np.uint32), np.array([0, P - 1] * 6, np.uint32), np.array([(P - 1) // 2] * 12, np.uint32)]
    for s in states:
        got = s.copy()
        rt.lib().zkir_poseidon2_permute(got.ctypes.data)
        assert np.array_equal(got, so.permute(s)), s
        scaled = s.copy()
        rt.lib().zkir_poseidon2_permute_scaled(scaled.ctypes.data, 1)               # the formulation the hash kernels run
        assert np.array_equal(scaled, got), s
    # iterate the permutation: 200 dependent applications explore states nobody picked
    s = np.arange(12, dtype=np.uint32)
    want = s.copy()
    chained = s.copy()
    rt.lib().zkir_poseidon2_permute_scaled(chained.ctypes.data, 200)                # with the sponge's carry step in between
    for _ in range(200):
        rt.lib().zkir_poseidon2_permute(s.ctypes.data)
        want = so.permute(want)
    assert np.array_equal(s, want)
    assert np.array_equal(chained, want)


@pytest.mark.parametrize("prog,n,cfg", [("fib_endless_program", 300, {}), ("compare_loop_program", 700, {}), ("call_loop_program", 500, {}), ("signed_loop_program", 700, {}),
                                        ("cmov_loop_program", 400, {}), ("sha256_chain_program", 300, {}), ("signed_loop_program", 200, {"enable_deferred_model": True}),
                                        ("cmov_loop_program", 200, {"enable_deferred_model": True}), ("fib_endless_program", 5, {})])
def test_main_trace_row_code_matches_oracle_on_the_host(prog, n, cfg):
    """The per-row code of main_trace_kernel is one host + device function (stark.hip: main_trace_row); zkir_main_trace_host runs it on the CPU
    over HOST trace columns.  Every committed column of every row — padding included — must equal the oracle's main trace: the kernel's logic
    is checked without a GPU (the GPU tests then check the same code where it ships)."""
    import numpy as np
    from oracle import api as oracle, stark_api as so
    blob = getattr(spec, prog)().to_bytes()
    rows = oracle.run(blob, max_cycles=n, enable_execution_trace=True, **cfg).rows
    deferred = bool(cfg)
    nr = len(rows)
    cyc, pc, ins = (np.ascontiguousarray(rows[f]) for f in ("cycle", "pc", "instruction"))
    regs, bb_, bt, bp, st = (np.ascontiguousarray(rows[f].T) for f in ("registers", "bound_bits", "bound_tag", "bound_payload", "reg_state"))
    tc = rt.TraceColumnsC(cyc.ctypes.data, pc.ctypes.data, ins.ctypes.data, regs.ctypes.data, bb_.ctypes.data, bt.ctypes.data, bp.ctypes.data, st.ctypes.data, nr)
    N = 1 << so.padded_log_n(nr)
    wm = so.committed_width(deferred)
    out = np.zeros((wm // 8, N, 8), np.uint32)
    L = rt.lib()
    L.zkir_main_trace_host.restype = C.c_int
    L.zkir_main_trace_host.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    assert L.zkir_main_trace_host(C.byref(tc), nr, int(deferred), out.ctypes.data) == 0
    got = out.transpose(0, 2, 1).reshape(wm, N)                                   # B8 blocks [N][8] -> [column][row]
    want = so.to_committed(so.main_trace(rows, so.public_inputs(nr, blob, deferred=deferred)), deferred)
    for k in range(wm):
        assert np.array_equal(got[k], want[k]), f"committed column {k}: first difference at row {int(np.nonzero(got[k] != want[k])[0][0])}"


def test_quotient_arithmetic_is_sound_on_the_constraint_list():
    """air::BoundOps runs air::eval on value bounds: every lazy value is reduced before it could overflow, every 96-bit sum stays below 2^73, every
    constraint index is pushed at most once — in both VM modes."""
    import ctypes as C
    L = rt.lib()
    L.zkir_air_check_bounds.restype = C.c_int
    L.zkir_air_check_bounds.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
    for mode in (0, 1, 2):                                   # default, deferred, default + the I/O argument
        why = C.create_string_buffer(256)
        assert L.zkir_air_check_bounds(mode, why, 256) == 0, why.value.decode()


@pytest.mark.parametrize("deferred", [False, True])
def test_quotient_evaluation_matches_oracle_constraints(deferred):
    """The constraint list as the quotient kernel evaluates it (QuotientOps on the host: lazy arithmetic, 96-bit sums, selector-wise accumulators, boundary
    constants folded out) against the oracle's naive constraints_sum, on random rows (no row of a real trace: every constraint is non-zero), random lookup
    parameters, selectors, boundary states and alpha — and on rows of all p - 1 (the largest words)."""
    import ctypes as C
    import numpy as np
    from oracle import stark_api as so
    L = rt.lib()
    L.zkir_air_eval_host.restype = None
    L.zkir_air_eval_host.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_void_p]
    P = so.P
    rng = np.random.default_rng(2026)
    virt = [9, 10, 11] + ([57] if deferred else list(range(57, 73)) + [161])       # air.h is_virtual: R0's limbs; the storage states (default: all 16 + class oj)
    blob = spec.fib_program(5).to_bytes()
    pub = so.public_inputs(64, blob, deferred=deferred)
    for trial in range(40):
        big = trial >= 36
        def words(n):
            return np.full(n, P - 1, np.uint32) if big else rng.integers(0, P, n).astype(np.uint32)
        loc, nxt, aloc, anxt, lk, first, last, alpha = words(172), words(172), words(40), words(40), words(56), words(68), words(68), words(4)
        loc[virt] = 0; nxt[virt] = 0
        sel = words(3)
        want = so.constraints_eval_states(loc, nxt, aloc, anxt, lk, sel[0], sel[1], sel[2], pub, first, last, alpha)
        got = np.zeros(4, np.uint32)
        L.zkir_air_eval_host(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(sel[0]), int(sel[1]), int(sel[2]),
                             first.ctypes.data, last.ctypes.data, alpha.ctypes.data, int(deferred), None, got.ctypes.data)
        assert np.array_equal(got, want), (trial, got, want)


def test_quotient_evaluation_matches_oracle_constraints_mode2():
    """The same in MODE 2 (default VM mode + the I/O argument: 180 logical / 48 aux columns, 430 constraints, the counters' boundary words, n_in among the lookup parameters)."""
    import ctypes as C
    import numpy as np
    from oracle import stark_api as so
    L = rt.lib()
    L.zkir_air_eval_host.restype = None
    L.zkir_air_eval_host.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_void_p]
    LO = so.lib()
    LO.so_constraints_eval_io.restype = C.c_int
    LO.so_constraints_eval_io.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_void_p] * 6
    P = so.P
    rng = np.random.default_rng(2027)
    virt = [9, 10, 11] + list(range(57, 73)) + [161]
    blob = spec.fib_program(5).to_bytes()
    pub = so.public_inputs(64, blob, [], [5], (1, 0), io_mode=True)
    assert LO.so_num_constraints_for(2) == 430
    for trial in range(40):
        big = trial >= 36
        def words(n):
            return np.full(n, P - 1, np.uint32) if big else rng.integers(0, P, n).astype(np.uint32)
        loc, nxt, aloc, anxt, lk, first, last, cnt, alpha, sel = words(180), words(180), words(48), words(48), words(57), words(68), words(68), words(4), words(4), words(3)
        loc[virt] = 0; nxt[virt] = 0
        want, got = np.zeros(4, np.uint32), np.zeros(4, np.uint32)
        LO.so_constraints_eval_io(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(sel[0]), int(sel[1]), int(sel[2]), C.byref(pub),
                                  first.ctypes.data, last.ctypes.data, cnt.ctypes.data, alpha.ctypes.data, want.ctypes.data)
        L.zkir_air_eval_host(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(sel[0]), int(sel[1]), int(sel[2]),
                             first.ctypes.data, last.ctypes.data, alpha.ctypes.data, 2, cnt.ctypes.data, got.ctypes.data)
        assert np.array_equal(got, want), (trial, got, want)


def _io_program():
    """READ, READ, WRITE their 40-bit sum, READ on the exhausted tape, WRITE that 0, EXIT(3) (syscall.rs:101-121)."""
    code = [spec.addi(10, 0, 1), spec.ecall(), spec.addi(5, 10, 0), spec.addi(10, 0, 1), spec.ecall(), spec.add(6, 5, 10), spec.addi(11, 6, 0), spec.addi(10, 0, 2), spec.ecall(),
            spec.addi(10, 0, 1), spec.ecall(), spec.addi(11, 10, 0), spec.addi(10, 0, 2), spec.ecall(), spec.addi(10, 0, 0), spec.addi(11, 0, 3), spec.ecall()]
    return spec.Program.from_code(code).to_bytes(), [1000, (1 << 45) + 77]


@pytest.mark.parametrize("which", ["io", "fib12", "sha"])
def test_main_trace_mode2_row_code_matches_oracle_on_the_host(which):
    """zkir_main_trace_io_host (stark.hip: main_trace_row<2> + the sequential form of the ecall prefix counts) against the oracle's mode-2 main trace: every committed column
    of every row, on a program that reads, writes and exhausts its tape, on fib(12) (one WRITE, exit) and on the SHA-256 chain (hash ecalls only)."""
    import numpy as np
    from oracle import api as oracle, stark_api as so
    if which == "io":
        blob, ins = _io_program()
        res = oracle.run(blob, ins, enable_execution_trace=True)
    elif which == "fib12":
        blob, ins = spec.fib_program(12).to_bytes(), []
        res = oracle.run(blob, enable_execution_trace=True)
    else:
        blob, ins = spec.sha256_chain_program().to_bytes(), []
        res = oracle.run(blob, max_cycles=300, enable_execution_trace=True)
    rows = res.rows
    nr = len(rows)
    cyc, pc, ins_c = (np.ascontiguousarray(rows[f]) for f in ("cycle", "pc", "instruction"))
    regs, bb_, bt, bp, st = (np.ascontiguousarray(rows[f].T) for f in ("registers", "bound_bits", "bound_tag", "bound_payload", "reg_state"))
    tc = rt.TraceColumnsC(cyc.ctypes.data, pc.ctypes.data, ins_c.ctypes.data, regs.ctypes.data, bb_.ctypes.data, bt.ctypes.data, bp.ctypes.data, st.ctypes.data, nr)
    N = 1 << so.padded_log_n(nr)
    wm = so.committed_width(2)
    assert wm == 160
    out = np.zeros((wm // 8, N, 8), np.uint32)
    tape = np.asarray(ins if ins else [0], dtype=np.uint64)

    class IoArgs(C.Structure):
        _fields_ = [("inputs", C.c_void_p), ("n_inputs", C.c_uint64), ("writes_before", C.c_uint64), ("reads_before", C.c_uint64)]
    io = IoArgs(tape.ctypes.data, len(ins), 0, 0)
    L = rt.lib()
    L.zkir_main_trace_io_host.restype = C.c_int
    L.zkir_main_trace_io_host.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    assert L.zkir_main_trace_io_host(C.byref(tc), nr, C.byref(io), out.ctypes.data) == 0
    got = out.transpose(0, 2, 1).reshape(wm, N)
    pub = so.public_inputs(nr, blob, ins, list(res.outputs), (res.halt_kind, res.halt_code), io_mode=True)
    want = so.to_committed(so.main_trace(rows, pub), 2)
    for k in range(wm):
        assert np.array_equal(got[k], want[k]), f"committed column {k}: first difference at row {int(np.nonzero(got[k] != want[k])[0][0])}"


# ---- MODE 3 (round 4): mode 2 + the memory argument — the product's host-side pieces against the oracle (no GPU) ------------------------------------------------
def test_air_bounds_mode3():
    """The quotient kernel's lazy arithmetic is sound on the mode-3 constraint list too (air::BoundOps on air::eval)."""
    import ctypes as C
    L = rt.lib()
    why = C.create_string_buffer(256)
    L.zkir_air_check_bounds.restype = C.c_int
    L.zkir_air_check_bounds.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
    assert L.zkir_air_check_bounds(3, why, 256) == 0, why.value.decode()


def test_quotient_evaluation_matches_oracle_constraints_mode3():
    """air::eval under the quotient kernel's arithmetic (host build) against the oracle's constraints_sum in MODE 3: 284 logical / 96 aux columns, 636 constraints."""
    import ctypes as C
    import numpy as np
    from oracle import stark_api as so
    L = rt.lib()
    L.zkir_air_eval_host.restype = None
    L.zkir_air_eval_host.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_void_p]
    LO = so.lib()
    LO.so_constraints_eval_io.restype = C.c_int
    LO.so_constraints_eval_io.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_void_p] * 6
    P = so.P
    rng = np.random.default_rng(2028)
    virt = [9, 10, 11] + list(range(57, 73)) + [161]
    blob = spec.fib_program(5).to_bytes()
    pub = so.public_inputs(64, blob, [], [5], (1, 0), mem_mode=True)
    assert LO.so_num_constraints_for(3) == 636
    for trial in range(40):
        big = trial >= 36
        def words(n):
            return np.full(n, P - 1, np.uint32) if big else rng.integers(0, P, n).astype(np.uint32)
        loc, nxt, aloc, anxt, lk, first, last, cnt, alpha, sel = words(284), words(284), words(96), words(96), words(57), words(68), words(68), words(4), words(4), words(3)
        loc[virt] = 0; nxt[virt] = 0
        want, got = np.zeros(4, np.uint32), np.zeros(4, np.uint32)
        LO.so_constraints_eval_io(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(sel[0]), int(sel[1]), int(sel[2]), C.byref(pub),
                                  first.ctypes.data, last.ctypes.data, cnt.ctypes.data, alpha.ctypes.data, want.ctypes.data)
        L.zkir_air_eval_host(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(sel[0]), int(sel[1]), int(sel[2]),
                             first.ctypes.data, last.ctypes.data, alpha.ctypes.data, 3, cnt.ctypes.data, got.ctypes.data)
        assert np.array_equal(got, want), (trial, got, want)


def _mode3_programs():
    import programs as pg
    out = {name: getattr(pg, name)() for name in ("mem_sw_lw", "timestamps", "loads_stores", "alu_all", "q9_access_at_own_pc", "echo5")}
    for seed in (1, 3, 5):
        blob, ins = pg.random_program(seed, hashes=False)
        out[f"random{seed}"] = (blob, ins, {})
    return out


@pytest.mark.parametrize("name", ["mem_sw_lw", "timestamps", "loads_stores", "alu_all", "q9_access_at_own_pc", "echo5", "random1", "random3", "random5"])
def test_memcheck_witness_and_main_trace_mode3_match_oracle_on_the_host(name):
    """zkir_memcheck_witness_of (the host's sequential memory replay over the delta log) yields the oracle's touched cells, and zkir_main_trace_mem_host (stark.hip:
    main_trace_row<3>) the oracle's mode-3 main trace: every committed column of every row — all ten loads / stores, sign extension, the program image, Q9's access at its own pc."""
    import ctypes as C
    import numpy as np
    from oracle import api as oracle, stark_api as so
    blob, ins, cfg = _mode3_programs()[name]
    cfg = {k: v for k, v in cfg.items() if k == "max_cycles"}
    res = oracle.run(blob, list(ins), enable_execution_trace=True, **cfg)
    rows, nr = res.rows, len(res.rows)
    log = rt.interpret(blob, list(ins), rt.VMConfig(enable_execution_trace=True, **cfg))
    assert log.n_rows == nr
    pub = rt.public_inputs(log, blob, list(ins), mem_mode=True, mem_witness="host")
    opub = so.public_inputs(nr, blob, list(ins), list(res.outputs), (res.halt_kind, res.halt_code), mem_mode=True)
    assert pub.deferred == 3 and list(pub.io_digest) == list(opub.io)
    want_cells = so.mem_cells(rows, opub)
    assert pub.n_cells == len(want_cells)
    if pub.n_cells:
        ca = np.ctypeslib.as_array(C.cast(pub.cell_addr, C.POINTER(C.c_uint64)), (pub.n_cells,))
        cb = np.ctypeslib.as_array(C.cast(pub.cell_bytes, C.POINTER(C.c_uint64)), (pub.n_cells,))
        ct = np.ctypeslib.as_array(C.cast(pub.cell_time, C.POINTER(C.c_uint32)), (pub.n_cells,))
    for k in range(pub.n_cells):
        w = [int(x) for x in want_cells[k]]
        assert (int(ca[k]) & 0xFFFFF, int(ca[k]) >> 20, int(ct[k]), [(int(cb[k]) >> (16 * i)) & 0xFFFF for i in range(4)]) == (w[0], w[1], w[2], w[3:])
    cyc, pc, ins_c = (np.ascontiguousarray(rows[f]) for f in ("cycle", "pc", "instruction"))
    regs, bb_, bt, bp, st = (np.ascontiguousarray(rows[f].T) for f in ("registers", "bound_bits", "bound_tag", "bound_payload", "reg_state"))
    tc = rt.TraceColumnsC(cyc.ctypes.data, pc.ctypes.data, ins_c.ctypes.data, regs.ctypes.data, bb_.ctypes.data, bt.ctypes.data, bp.ctypes.data, st.ctypes.data, nr)
    N = 1 << so.padded_log_n(nr)
    wm = so.committed_width(3)
    assert wm == 264
    out = np.zeros((wm // 8, N, 8), np.uint32)
    tape = np.asarray(list(ins) if len(ins) else [0], dtype=np.uint64)

    class IoArgs(C.Structure):
        _fields_ = [("inputs", C.c_void_p), ("n_inputs", C.c_uint64), ("writes_before", C.c_uint64), ("reads_before", C.c_uint64)]
    io = IoArgs(tape.ctypes.data, len(ins), 0, 0)
    L = rt.lib()
    L.zkir_main_trace_mem_host.restype = C.c_int
    L.zkir_main_trace_mem_host.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.zkir_main_trace_mem_host(C.byref(tc), nr, C.byref(io), pub.mem_old, pub.mem_told, out.ctypes.data) == 0
    got = out.transpose(0, 2, 1).reshape(wm, N)
    want = so.to_committed(so.main_trace(rows, opub), 3)
    for k in range(wm):
        assert np.array_equal(got[k], want[k]), f"committed column {k}: first difference at row {int(np.nonzero(got[k] != want[k])[0][0])}"


# ---- MODE 4 (round 6): mode 3 + the wide-arithmetic class MULH / DIVU / REMU / DIV / REM — the product's host-side pieces against the oracle (no GPU) ------------------------
def test_air_bounds_mode4():
    """The quotient kernel's lazy arithmetic is sound on the mode-4 constraint list (712 constraints) too (air::BoundOps on air::eval)."""
    import ctypes as C
    L = rt.lib()
    why = C.create_string_buffer(256)
    L.zkir_air_check_bounds.restype = C.c_int
    L.zkir_air_check_bounds.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
    assert L.zkir_air_check_bounds(4, why, 256) == 0, why.value.decode()


def test_quotient_evaluation_matches_oracle_constraints_mode4():
    """air::eval under the quotient kernel's arithmetic (host build) against the oracle's constraints_sum in MODE 4: 308 logical / 128 aux columns, 712 constraints (the
    boundary cell's address enters through two lookup parameters, LK_B0 / LK_B1: the oracle derives it from the program, here fib(5)'s 19 code words: code_size % 8 == 4)."""
    import ctypes as C
    import numpy as np
    from oracle import stark_api as so
    L = rt.lib()
    L.zkir_air_eval_host.restype = None
    L.zkir_air_eval_host.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_void_p] * 3 + [C.c_uint32, C.c_void_p, C.c_void_p]
    LO = so.lib()
    LO.so_constraints_eval_io.restype = C.c_int
    LO.so_constraints_eval_io.argtypes = [C.c_void_p] * 5 + [C.c_uint32] * 3 + [C.c_void_p] * 6
    P = so.P
    rng = np.random.default_rng(2029)
    virt = [9, 10, 11] + list(range(57, 73)) + [161]
    blob = spec.fib_program(5).to_bytes()
    pub = so.public_inputs(64, blob, [], [5], (1, 0), wide_mode=True)
    assert LO.so_num_constraints_for(4) == 712 and so.logical_width(4) == 308 and so.aux_width(4) == 128 and so.committed_width(4) == 288
    code_size = int.from_bytes(blob[16:20], "little")
    bcell = 0x1000 + code_size - 4 if code_size % 8 == 4 else 0x1000
    for trial in range(40):
        big = trial >= 36
        def words(n):
            return np.full(n, P - 1, np.uint32) if big else rng.integers(0, P, n).astype(np.uint32)
        loc, nxt, aloc, anxt, lk, first, last, cnt, alpha, sel = words(308), words(308), words(128), words(128), words(59), words(68), words(68), words(4), words(4), words(3)
        loc[virt] = 0; nxt[virt] = 0
        lk[57], lk[58] = bcell & 0xFFFFF, (bcell >> 20) & 0xFFFFF
        want, got = np.zeros(4, np.uint32), np.zeros(4, np.uint32)
        LO.so_constraints_eval_io(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(sel[0]), int(sel[1]), int(sel[2]), C.byref(pub),
                                  first.ctypes.data, last.ctypes.data, cnt.ctypes.data, alpha.ctypes.data, want.ctypes.data)
        L.zkir_air_eval_host(loc.ctypes.data, nxt.ctypes.data, aloc.ctypes.data, anxt.ctypes.data, lk.ctypes.data, int(sel[0]), int(sel[1]), int(sel[2]),
                             first.ctypes.data, last.ctypes.data, alpha.ctypes.data, 4, cnt.ctypes.data, got.ctypes.data)
        assert np.array_equal(got, want), (trial, got, want)


@pytest.mark.parametrize("name", ["wide_grid", "alu_all", "loads_stores", "random3", "sha_chain", "sha256_hello"])
def test_main_trace_mode4_matches_oracle_on_the_host(name):
    """zkir_main_trace_wide_host (stark.hip: main_trace_row<4>) writes the oracle's mode-4 main trace: every committed column of every row — the five wide opcodes over a grid
    of operands (the largest carries, dividend < divisor, equal operands, rd = r0), and the mode-3 columns unchanged on programs that use none of them."""
    import ctypes as C
    import numpy as np
    import programs as pg
    from oracle import api as oracle, stark_api as so
    if name.startswith("random"):
        blob, ins = pg.random_program(int(name[6:]), hashes=False); cfg = {}
    elif name == "sha_chain":
        blob, ins, cfg = spec.sha256_chain_program().to_bytes(), [], {"max_cycles": 600}
    else:
        blob, ins, cfg = getattr(pg, name)()
    cfg = {k: v for k, v in cfg.items() if k == "max_cycles"}
    res = oracle.run(blob, list(ins), enable_execution_trace=True, **cfg)
    rows, nr = res.rows, len(res.rows)
    log = rt.interpret(blob, list(ins), rt.VMConfig(enable_execution_trace=True, **cfg))
    pub = rt.public_inputs(log, blob, list(ins), wide_mode=True, mem_witness="host")
    opub = so.public_inputs(nr, blob, list(ins), list(res.outputs), (res.halt_kind, res.halt_code), wide_mode=True)
    assert pub.deferred == 4 and list(pub.io_digest) == list(opub.io)
    # the hash tape of the host witness (zkir_memcheck_witness_of_mode, hashcall.h) is the oracle's hash section word for word, the touched cells its touched cells
    want_hash = so.hash_section(rows, opub)
    got_hash = np.ctypeslib.as_array(C.cast(pub.hash_section, C.POINTER(C.c_uint32)), (pub.hash_section_words,)) if pub.hash_section_words else np.zeros(1, np.uint32)
    assert np.array_equal(got_hash, want_hash) and (int(want_hash[0]) > 0) == (name in ("sha_chain", "sha256_hello"))
    assert pub.n_cells == len(so.mem_cells(rows, opub))
    cyc, pc, ins_c = (np.ascontiguousarray(rows[f]) for f in ("cycle", "pc", "instruction"))
    regs, bb_, bt, bp, st = (np.ascontiguousarray(rows[f].T) for f in ("registers", "bound_bits", "bound_tag", "bound_payload", "reg_state"))
    tc = rt.TraceColumnsC(cyc.ctypes.data, pc.ctypes.data, ins_c.ctypes.data, regs.ctypes.data, bb_.ctypes.data, bt.ctypes.data, bp.ctypes.data, st.ctypes.data, nr)
    N = 1 << so.padded_log_n(nr)
    wm = so.committed_width(4)
    out = np.zeros((wm // 8, N, 8), np.uint32)
    tape = np.asarray(list(ins) if len(ins) else [0], dtype=np.uint64)

    class IoArgs(C.Structure):
        _fields_ = [("inputs", C.c_void_p), ("n_inputs", C.c_uint64), ("writes_before", C.c_uint64), ("reads_before", C.c_uint64)]
    io = IoArgs(tape.ctypes.data, len(ins), 0, 0)
    L = rt.lib()
    L.zkir_main_trace_wide_host.restype = C.c_int
    L.zkir_main_trace_wide_host.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    assert L.zkir_main_trace_wide_host(C.byref(tc), nr, C.byref(io), pub.mem_old, pub.mem_told, int.from_bytes(blob[16:20], "little"), out.ctypes.data) == 0
    got = out.transpose(0, 2, 1).reshape(wm, N)
    want = so.to_committed(so.main_trace(rows, opub), 4)
    for k in range(wm):
        assert np.array_equal(got[k], want[k]), f"committed column {k}: first difference at row {int(np.nonzero(got[k] != want[k])[0][0])}"


def test_memcheck_witness_refuses_runs_outside_the_air():
    import programs as pg
    blob, ins, _ = pg.sha256_hello()
    log = rt.interpret(blob, list(ins), rt.VMConfig(enable_execution_trace=True))
    with pytest.raises(Exception, match="hash syscall"):
        rt.public_inputs(log, blob, list(ins), mem_mode=True, mem_witness="host")
    O, E = spec.Opcode, spec.encode                                      # LB sign-extends 0x80 to 64 bits (Q1): the next load's address is 0xFFFF_FFFF_FFFF_FF80
    blob = pg._p([pg.A(5, 0, 0x4000), pg.A(1, 0, 0x80), E(O.SB, rs1=5, rs2=1, imm=0), E(O.LB, 1, 5, imm=0), spec.lw(2, 1, 0), pg.EB])
    log = rt.interpret(blob, [], rt.VMConfig(enable_execution_trace=True))
    with pytest.raises(Exception, match="2\\^40"):
        rt.public_inputs(log, blob, [], mem_mode=True, mem_witness="host")
    blob, ins, _ = pg.fib30()
    log = rt.interpret(blob, list(ins), rt.VMConfig(enable_execution_trace=True))
    shard = log.shard(10, 20) if hasattr(log, "shard") else None
    if shard is not None:
        with pytest.raises(Exception, match="whole run"):
            rt.MemcheckWitness(shard, blob)
