#!/usr/bin/env python3
"""Writes tests/golden/config_proofs.json: the COMPLETE proof (format v10, mode 0) of the fib_endless run halted at 2^k cycles — k = 20 is the run BASELINE's metric
"end-to-end prove ms, 2^20-cycle fib" is quoted on — computed by the CPU ORACLE alone (oracle/zkir_oracle.cpp executes, oracle/stark_oracle.cpp proves: textbook
arithmetic, one thread, ~13 minutes and ~5 GB at 2^20).  Kept per size: the proof's length in words, the SHA-256 of its little-endian words, and 257 evenly spaced
words (so that a mismatch says roughly WHERE).  tests/test_gpu_large.py::test_proof_equals_the_oracles_at_config_size compares the GPU prover's proof with it:
"proof bytes bit-identical" at the size the metric names, where tests/test_gpu_stark.py compares whole proofs up to 2^13 rows.

What this pins: nothing outside this repository — the prover is self-defined (the reference has none): parity stays UNPINNED vs seceq/zkir.

Does not import the product.  Run: python tests/golden/make_config_proofs.py [log2_rows ...]   (default 16 18 20)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from make_config_roots import FIB_ENDLESS  # noqa: E402  (the same program blob, own encoder)
from oracle import api as oracle, stark_api as so  # noqa: E402  (test infrastructure only)

N_SAMPLES = 257


def sample_positions(n_words: int):
    return [int(i * (n_words - 1) // (N_SAMPLES - 1)) for i in range(N_SAMPLES)]


def main():
    ks = [int(a) for a in sys.argv[1:]] or [16, 18, 20]
    path = os.path.join(HERE, "config_proofs.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["_about"] = ("whole proofs (ZKIR-STARK format v10, mode 0) of the fib_endless run halted at 2^k cycles, written by tests/golden/make_config_proofs.py from the CPU "
                     "oracle alone: length, SHA-256 of the little-endian u32 words, 257 evenly spaced words; self-defined prover: parity UNPINNED vs seceq/zkir (it has "
                     "no prover) — the file makes GPU proof == independent CPU proof literal at the size BASELINE's metric names (2^20 cycles)")
    out["program_blob_hex"] = FIB_ENDLESS.hex()
    proofs = out.setdefault("proofs", {})
    for k in ks:
        t0 = time.time()
        res = oracle.run(FIB_ENDLESS, max_cycles=1 << k, enable_execution_trace=True)
        pub = so.public_inputs(len(res.rows), FIB_ENDLESS, [], list(res.outputs), (res.halt_kind, res.halt_code))
        proof = np.ascontiguousarray(so.prove(res.rows, pub), dtype="<u4")
        assert so.verify(proof, pub) == 0
        pos = sample_positions(len(proof))
        proofs[str(k)] = {"rows": 1 << k, "words": int(len(proof)), "sha256": hashlib.sha256(proof.tobytes()).hexdigest(),
                          "samples": [int(proof[p]) for p in pos], "oracle_seconds": round(time.time() - t0, 1)}
        print(k, proofs[str(k)]["words"], proofs[str(k)]["sha256"], proofs[str(k)]["oracle_seconds"], flush=True)
        cur = json.load(open(path)) if os.path.exists(path) else {}          # (several sizes may be running side by side: merge, do not overwrite)
        cur.update({"_about": out["_about"], "program_blob_hex": out["program_blob_hex"]})
        cur.setdefault("proofs", {})[str(k)] = proofs[str(k)]
        json.dump(cur, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
