#!/usr/bin/env python3
"""VALU utilisation per kernel from a rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass.

usage: extract_valu.py <dir with *.db> <out.txt>
VALUBusy (rocprof's derived metric) = 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / (GRBM_GUI_ACTIVE / XCDs): the share of SIMD issue
cycles spent executing vector-ALU instructions — the roofline of the integer-ALU-bound Poseidon2 kernels.
"""
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SIMD_NUM = 256 * 4
XCDS = 8          # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (sum / wall time = 8 x ~2.3 GHz)


def main():
    f = sorted(glob.glob(sys.argv[1] + "/*.db"))
    if not f:
        print("no counter db"); return
    c = sqlite3.connect(f[0])
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    per = {}
    for name, ctr, v, n, dur in rows:
        s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        per.setdefault(s, {"launches": n, "avg_ns": dur})[ctr] = v
    from zkir_amd.build import sources_sha16
    lines = [f"# kernels_sha16: {sources_sha16()}", "# kernel, launches, avg_us, SQ_INSTS_VALU (wave instr), SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE, VALUBusy % = 100*ACTIVE*4/SIMDs/(GUI_ACTIVE/8 XCDs), VALU instr per SIMD-cycle"]
    for s, d in sorted(per.items(), key=lambda kv: -kv[1].get("avg_ns", 0) * kv[1]["launches"]):
        gui, act, insts = d.get("GRBM_GUI_ACTIVE"), d.get("SQ_ACTIVE_INST_VALU"), d.get("SQ_INSTS_VALU")
        busy = 100.0 * act * 4 / SIMD_NUM / (gui / XCDS) if gui and act else float("nan")
        ipc = insts / SIMD_NUM / (gui / XCDS) if gui and insts else float("nan")
        lines.append(f"{s:60s} {d['launches']:5d} {d['avg_ns'] / 1e3:10.1f} {insts or 0:14.0f} {act or 0:14.0f} {gui or 0:12.0f} {busy:8.1f} {ipc:8.4f}")
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
