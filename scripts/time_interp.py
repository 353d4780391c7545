#!/usr/bin/env python3
"""Host interpreter rates on this box: traced, untraced, and a trace window (fast-forward + traced tail)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkir_amd import runtime as rt, spec
blob = spec.fib_endless_program().to_bytes()
n = 1 << 23
def t(f):
    best = 1e9
    for _ in range(3):
        a = time.perf_counter(); l = f(); best = min(best, time.perf_counter() - a); l.close()
    return best
tr = t(lambda: rt.interpret(blob, [], rt.VMConfig(max_cycles=n, enable_execution_trace=True)))
un = t(lambda: rt.interpret(blob, [], rt.VMConfig(max_cycles=n)))
w1 = t(lambda: rt.interpret(blob, [], rt.VMConfig(max_cycles=2 * n, enable_execution_trace=True), window=(n, 2 * n)))
w7 = t(lambda: rt.interpret(blob, [], rt.VMConfig(max_cycles=8 * n, enable_execution_trace=True), window=(7 * n, 8 * n)))
print(f"traced {tr / n * 1e9:.2f} ns/row, untraced {un / n * 1e9:.2f} ns/row, window [n,2n) {w1 * 1e3:.1f} ms (fast-forward {(w1 - tr) / n * 1e9:.2f} ns/row), window [7n,8n) {w7 * 1e3:.1f} ms (fast-forward {(w7 - tr) / (7 * n) * 1e9:.2f} ns/row)")
