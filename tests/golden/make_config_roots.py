#!/usr/bin/env python3
"""Writes tests/golden/config_roots.json: the trace-commitment root of BASELINE configs[1] (2^20-cycle fib, 152 committed columns, blow-up 2) and of
smaller sizes of the same run, computed by the CPU ORACLE alone (oracle/zkir_oracle.cpp executes, oracle/stark_oracle.cpp commits: textbook
arithmetic, one thread).  tests/test_gpu_large.py compares the GPU's root with it word for word: "bit-exact root vs CPU" at the size the config names.

What this pins: nothing outside this repository — the commit stage is self-defined (the reference has no prover); the file makes the GPU
and the independent CPU statement of the same spec agree AT FULL SIZE, where the test suite otherwise samples.

Does not import the product.  Run: python tests/golden/make_config_roots.py [max_log2_rows=20]   (2^20 rows: ~4 minutes, ~3 GB; 2^24 rows, column-blocked on 8 threads: ~15 minutes, ~25 GB)
     python tests/golden/make_config_roots.py sharded 26 8   BASELINE configs[3]'s own workload: the 2^26-cycle run cut into 8 row shards of 2^23, each shard
                                                             committed on its own (what one GPU of the 8-GPU run commits), the 8 subtree roots capped by three
                                                             levels of 2-to-1 compressions (what every rank computes after the all-gather) -> roots["26x8"]
"""
import json
import os
import struct
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import api as oracle, stark_api as so  # noqa: E402  (test infrastructure only)

ADD, ADDI, BNE, JAL = 0x00, 0x08, 0x41, 0x48


def r_(op, rd, rs1, rs2): return op | rd << 7 | rs1 << 11 | rs2 << 15
def i_(op, rd, rs1, imm): return op | rd << 7 | rs1 << 11 | (imm & 0x1FFFF) << 15
def j_(op, rd, off): return op | rd << 7 | (off & 0x1FFFFF) << 11


def blob(code):
    hdr = struct.pack("<IIBBBBIIIII", 0x52494B5A, 0x00030004, 20, 2, 2, 0, 0x1000, 4 * len(code), 0, 0, 1 << 20)
    return hdr + b"".join(struct.pack("<I", w & 0xFFFFFFFF) for w in code)


# the v3.4 fib loop of tests/cross_module.rs:145-164 with a back edge instead of the exit (configs 2-4), halted by max_cycles
FIB_LOOP = [r_(ADD, 4, 1, 2), i_(ADDI, 1, 2, 0), i_(ADDI, 2, 4, 0), i_(ADDI, 3, 3, -1), i_(BNE, 3, 0, -16)]
FIB_ENDLESS = blob([i_(ADDI, 1, 0, 0), i_(ADDI, 2, 0, 1), i_(ADDI, 3, 0, 0)] + FIB_LOOP + [j_(JAL, 0, -20)])


def sharded(k, G):
    """Row shards [g n, (g+1) n), n = 2^k / G, of the 2^k-cycle run: the oracle re-executes the run keeping one shard's rows at a time (window mode),
    commits the shard exactly as a rank of `bench.py --gpus G` does (its own interpolants over 2^(k - log2 G) rows, its own Merkle subtree) and caps the roots."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_roots.json")
    out = json.load(open(path))
    key = f"{k}x{G}"
    n = (1 << k) // G
    threads = int(os.environ.get("ZKIR_ORACLE_THREADS", os.cpu_count() or 1))
    entry = out["roots"].get(key) or {"rows": 1 << k, "shards": G, "rows_per_shard": n, "shard_roots": [], "oracle_seconds": 0.0, "threads": threads}
    out["roots"][key] = entry
    for g in range(len(entry["shard_roots"]), G):
        t0 = time.time()
        rows = oracle.run(FIB_ENDLESS, max_cycles=1 << k, enable_execution_trace=True, keep_rows=(g * n, (g + 1) * n), want_sorted=False).rows
        assert len(rows) == n and int(rows["cycle"][0]) == g * n
        root = so.commit_trace_blocked(rows, 1, threads=threads)
        del rows
        entry["shard_roots"].append([int(x) for x in root])
        entry["oracle_seconds"] = round(entry["oracle_seconds"] + time.time() - t0, 1)
        print(key, g, entry["shard_roots"][-1], round(time.time() - t0, 1), flush=True)
        json.dump(out, open(path, "w"), indent=1)
    level = [[int(x) for x in r] for r in entry["shard_roots"]]
    while len(level) > 1:
        level = [[int(x) for x in so.compress(level[i], level[i + 1])] for i in range(0, len(level), 2)]
    entry["root"] = level[0]
    print(key, "capped root", entry["root"], flush=True)
    json.dump(out, open(path, "w"), indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "sharded":
        return sharded(int(sys.argv[2]), int(sys.argv[3]))
    kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_roots.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["_about"] = ("trace-commitment roots (Poseidon2-12 Merkle over the blow-up-2 coset LDE of the 152 committed main-trace columns) of the fib_endless run halted at "
                     "2^k cycles, written by tests/golden/make_config_roots.py from the CPU oracle alone; self-defined stages: parity UNPINNED vs seceq/zkir "
                     "(it has no prover) — the file makes GPU == independent CPU statement literal at the size BASELINE configs[1] names")
    out["program_blob_hex"] = FIB_ENDLESS.hex()
    roots = out.setdefault("roots", {})
    for k in sorted(set([12, 16, 18, kmax])):
        if str(k) in roots:
            continue
        t0 = time.time()
        rows = oracle.run(FIB_ENDLESS, max_cycles=1 << k, enable_execution_trace=True).rows
        if k >= 22:       # column-blocked (eight columns at a time, one sponge state per leaf: so_commit_trace_blocked — the same functions in the same order per leaf;
                          # tests/test_stark_oracle.py holds it equal to commit_trace); threads share columns / leaves, the arithmetic is untouched
            threads = os.cpu_count() or 1
            root = so.commit_trace_blocked(rows, 1, threads=threads)
        else:
            threads = 1
            root = so.commit_trace(rows, 1)
        roots[str(k)] = {"rows": 1 << k, "root": [int(x) for x in root], "oracle_seconds": round(time.time() - t0, 1), "threads": threads}
        print(k, roots[str(k)], flush=True)
        json.dump(out, open(path, "w"), indent=1)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
