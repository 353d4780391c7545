#!/usr/bin/env python3
"""One table of current numbers per kernel (VERDICT r4 task 9) out of the round's committed profile files:
    python scripts/kernel_table.py r05 > profiles/r05_kernel_table.md
time: rocprofv3 --kernel-trace --stats averages (profiles/<tag>_config1_kernel_stats.txt = the bench step, <tag>_prove_kernel_stats_2p20.txt = three proofs at 2^20 rows);
HBM bytes per launch: FETCH_SIZE / WRITE_SIZE passes (<tag>_*_pmc_traffic.json: FETCH doubled for 16-byte coalesced reads, MI355X_MICROARCH.md); VALU busy: <tag>_*_valu_busy.txt;
registers / spills / scratch: the shipped code object's metadata (<tag>_isa_hist.json, scripts/isa_hist.py)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")


def stats(path):
    out = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+) us\s+([\d.]+) us\s+([\d.]+)%", line)
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out


def busy(path):
    out = {}
    for line in open(path):
        if line.startswith("#") or len(line) < 70:
            continue
        f = line[60:].split()
        try:
            out[line[:60].strip()] = float(f[5])
        except (ValueError, IndexError):
            pass
    return out


def traffic(path):
    d = json.load(open(path))
    return {k: v["hbm_bytes_per_launch"] for k, v in d.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}


def regs():
    d = json.load(open(P("isa_hist.json")))
    return {k: v["regs"] for k, v in d.items() if isinstance(v, dict) and "regs" in v}


def key_of(name, table):
    base = name.split("<")[0]
    if name in table:
        return name
    cands = [k for k in table if k.split("<")[0] == base]
    return cands[0] if len(cands) == 1 else None


def emit(title, st, bz, tr, rg, per):
    print(f"**{title}**\n")
    print("| kernel | launches | avg µs | HBM bytes / launch (PMC) | GB/s | of 8 TB/s | VALU busy % | VGPR | spilled VGPR / SGPR | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, (calls, avg) in sorted(st.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        if name.startswith(("__amd", "void at::", "copy16", "modmul", "powers", "selector")):
            continue
        t = tr.get(key_of(name, tr) or "")
        b = bz.get(key_of(name, bz) or "")
        r = rg.get(key_of(name, rg) or "")
        gbs = t / (avg * 1e-6) / 1e9 if t else None
        print(f"| `{name}` | {calls / per:g} | {avg:.1f} | {t / 1e6:.1f} MB |" .replace("None", "—") if t else f"| `{name}` | {calls / per:g} | {avg:.1f} | — |", end="")
        print(f" {gbs:.0f} | {gbs / 8000:.3f} |" if gbs else " — | — |", end="")
        print(f" {b:.1f} |" if b is not None else " — |", end="")
        print(f" {r['vgpr']} | {r['vgpr_spill']} / {r['sgpr_spill']} | {r['scratch']} |" if r else " — | — | — |")
    print()


rg = regs()
c1 = stats(P("config1_kernel_stats.txt"))
steps = max(v[0] for k, v in c1.items() if k.startswith("leaf_hash"))
emit(f"The bench step (configs[1]: 2²⁰ rows, 152 columns; `profiles/{tag}_config1_kernel_stats.txt`, launches per step)", c1, busy(P("bench_commit_valu_busy.txt")), traffic(P("bench_commit_pmc_traffic.json")), rg, steps)
pv = stats(P("prove_kernel_stats_2p20.txt"))
emit(f"One proof at 2²⁰ rows, mode 0 (`profiles/{tag}_prove_kernel_stats_2p20.txt`: three proofs + setup; launches per proof)", pv, busy(P("prove_valu_busy_2p20.txt")), traffic(P("prove_2p20_pmc_traffic.json")), rg, 3)
