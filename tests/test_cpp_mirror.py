"""The C++ host mirror of the reference API (include/zkir_amd.hpp) and its re-statement of the reference's tests
(tests/cpp/reference_tests.cpp): compiled with g++ against the in-tree library; the default-config tests run on the host
interpreter alone, the execution-trace tests need the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "zkir_amd")
BIN = os.path.join(ROOT, "tests", "cpp", "reference_tests")


def _build():
    from zkir_amd import build as zbuild
    zbuild.build()
    src = os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp")
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "zkir_amd.hpp")),
                                                             os.path.getmtime(os.path.join(LIBDIR, "libzkir_amd.so"))):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-L", LIBDIR, "-lzkir_amd",
                               f"-Wl,-rpath,{LIBDIR}", "-o", BIN])


def _run(mode):
    import torch  # noqa: F401  (the library's HIP runtime is the one PyTorch bundles: make its directory visible to the loader)
    env = dict(os.environ)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env["LD_LIBRARY_PATH"] = torch_lib + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([BIN, mode], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_cpp_mirror_host_tests():
    _build()
    out = _run("host")
    assert "17 tests, 0 failures" in out, out


@pytest.mark.gpu
def test_cpp_mirror_gpu_tests():
    _build()
    out = _run("gpu")
    assert "27 tests, 0 failures" in out, out
