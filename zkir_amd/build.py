"""Build libzkir_amd.so (host interpreter + HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with gpurun.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzkir_amd.so")
OBJ = os.path.join(HERE, "build")

HOST_SOURCES = ["interp.cpp", "hashes.cpp", "verify.cpp"]
HIP_SOURCES = ["trace_fill.hip", "witness.hip", "ntt.hip", "stark.hip", "memcheck.hip", "abi.hip"]
HEADERS = ["host.h", "hashcall.h", "babybear.h", "poseidon2.h", "air.h", "stark_prove.inl", os.path.join("..", "..", "include", "zkir_amd.h"), os.path.join("..", "..", "include", "zkir_amd_experimental.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Field-arithmetic kernels are long chains of dependent integer ops with 12 independent chains per Poseidon2 layer: the
# max-ILP scheduler interleaves them (same instruction count, same occupancy), which is what the latency-bound kernels
# (Merkle tree tails, FRI layers) need and costs the throughput-bound ones nothing.
EXTRA = {"stark.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def sources_sha16() -> str:
    """sha256 prefix over the kernel sources the library is built from: the counter files under profiles/ carry the value they were measured at, so a
    reader without git (the GPU box) can tell whether they describe THIS build (bench.py roofline.traffic_source_freshness)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(HIP_SOURCES + [x for x in HEADERS if not x.startswith("..")]):      # the KERNEL sources (.hip and what they include): host-only files do not move a counter
        h.update(name.encode())
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def _newer(src: str, dst: str, deps) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + list(deps))


def build_variant(tag: str, defines, csrc: str = CSRC) -> str:
    """A second build of the library with extra -D flags (kernel experiments), as zkir_amd/variants/libzkir_amd_<tag>.so; run it
    with ZKIR_AMD_LIB=<that path>.  `csrc` may point at another checkout's sources (a baseline to time against)."""
    vdir = os.path.join(HERE, "variants", tag)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for name in HOST_SOURCES + HIP_SOURCES:
        src = os.path.join(csrc, name)
        obj = os.path.join(vdir, name + ".o")
        objs.append(obj)
        if name.endswith(".hip"):
            cmd = [HIPCC, f"--offload-arch={ARCH}", *COMMON, *EXTRA.get(name, []), *defines, "-c", src, "-o", obj]
        else:
            cmd = [HIPCC, "-x", "c++", *COMMON, "-march=x86-64-v2", *defines, "-c", src, "-o", obj]
        subprocess.check_call(cmd)
    out = os.path.join(HERE, "variants", f"libzkir_amd_{tag}.so")
    subprocess.check_call([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out, *objs])
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs, cmds = [], []
    for name in HOST_SOURCES + HIP_SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, name + ".o")
        objs.append(obj)
        if force or _newer(src, obj, deps):
            if name.endswith(".hip"):
                cmds.append([HIPCC, f"--offload-arch={ARCH}", *COMMON, *EXTRA.get(name, []), "-c", src, "-o", obj])
            else:
                cmds.append([HIPCC, "-x", "c++", *COMMON, "-march=x86-64-v2", "-c", src, "-o", obj])
    if cmds:                                                      # the translation units are independent: compile them side by side (stark.hip alone is two of the 2.5 minutes)
        from concurrent.futures import ThreadPoolExecutor
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=int(os.environ.get("ZKIR_BUILD_JOBS", "4"))) as ex:
            list(ex.map(run, cmds))
    if force or any(_newer(o, OUT, []) for o in objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
