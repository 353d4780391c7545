"""The C-ABI library loads and exports every symbol include/zkir_amd.h declares; host-only entry points
behave (error reporting, argument checks).  No compute on a device here."""
import ctypes as C
import os
import re

import pytest

from zkir_amd import runtime as rt, spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "zkir_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkir_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    L = rt.lib()
    names = _declared_functions()
    assert len(names) > 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/zkir_amd.h but not exported: {missing}"


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
