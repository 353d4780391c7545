"""zkir_amd — MI355X-native execution-trace path for ZKIR v3.4.

`zkir_amd.spec` is pure host logic (program blobs).  `zkir_amd.runtime` mirrors the reference's
`zkir_runtime` API (VM, VMConfig, ExecutionResult) on top of the C-ABI in include/zkir_amd.h and
fails loudly if the HIP library is not built.
"""
from . import spec  # noqa: F401
