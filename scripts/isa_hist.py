#!/usr/bin/env python3
"""Instruction histogram of the gfx950 kernels in the shipped library (VERDICT r4 task 1 / weak #5, #9).

    python scripts/isa_hist.py [--so zkir_amd/libzkir_amd.so] [--kernels leaf_hash,subtree,...] [--json profiles/r05_isa_hist.json] > profiles/r05_isa_hist.txt

Steps: copy the `.hip_fatbin` section out of the .so (llvm-objcopy), walk its clang offload bundles (`__CLANG_OFFLOAD_BUNDLE__`,
u64 count, then per entry u64 offset / u64 size / u64 triple length / triple), keep the gfx950 code objects, `llvm-objdump -d` each,
and count mnemonics per kernel symbol.  `llvm-readelf --notes` gives VGPR / SGPR / spill / scratch per kernel.

The per-class issue costs are the measured ones of profiles/r02_ubench_alu.txt (SIMD-cycles per wave64 instruction at 2.4 GHz):
the ALU bound of a kernel is  sum_i n_i c_i  SIMD-cycles per wave, so measured cycles / that sum <= 1 by construction when the
costs are right.  STATIC counts: a loop body is counted once — the `trip` table below gives the trip counts of the loops of the
kernels the bench line prices (checked against SQ_INSTS_VALU in the committed counter passes).
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# SIMD-cycles per wave64 instruction, profiles/r02_ubench_alu.txt (gfx950, 2.40 GHz): "full-rate" 32-bit ops ~2.3, everything else ~4.2,
# v_mad_u64_u32 4.48, v_cndmask (with a VCC dependency chain in the ubench) priced at the 4.2 of its class: the ubench's 20.6 measures a
# serialised vcc chain, not the instruction.
FULL_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32", "v_add_f32", "v_mul_f32",
             "v_sub_f32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_accvgpr_mov_b32", "v_nop"}
COST_SPECIAL = {"v_mad_u64_u32": 4.48, "v_mad_i64_i32": 4.48, "v_mul_lo_u32": 4.21, "v_mul_hi_u32": 4.09, "v_lshl_add_u64": 4.15}
COST_FULL, COST_OTHER = 2.3, 4.2
# The ARCHITECTURAL issue costs the bench line's ALU bound is built from: a SIMD has 16 lanes, so a wave64 instruction occupies it for 4 cycles; the
# full-rate 32-bit set above goes through in 2 (the ubench's 2.2-2.4 / 4.1-4.5 are these at the clock the ubench really ran at, plus loop overhead).
# They are LOWER bounds of every measured cost, so  sum n_i c_i  never exceeds the cycles a kernel really takes: frac <= 1 by construction.
ARCH_FULL, ARCH_OTHER = 2.0, 4.0


def arch_cost_of(mn: str) -> float:
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    return ARCH_FULL if base in FULL_RATE else ARCH_OTHER


def cost_of(mn: str) -> float:
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    if base in COST_SPECIAL:
        return COST_SPECIAL[base]
    return COST_FULL if base in FULL_RATE else COST_OTHER


def klass(mn: str) -> str:
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", mn)
    if base in ("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_mul_i32_i24"):
        return "multiplier"
    if base == "v_lshl_add_u64":
        return "add64"
    if base in ("v_mov_b32", "v_mov_b64", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_accvgpr_mov_b32"):
        return "copy"
    return "full_rate_32" if base in FULL_RATE else "other_valu"


def code_objects(so: str, workdir: str):
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat])
    blob = open(fat, "rb").read()
    outs, pos, idx = [], 0, 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            break
        (cnt,) = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(cnt):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                path = os.path.join(workdir, f"co{idx}.elf")
                open(path, "wb").write(blob[pos + off:pos + off + size])
                outs.append(path)
                idx += 1
        pos += len(MAGIC)
    return outs


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def notes(co: str):
    """kernel symbol -> {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, wg} from the AMDGPU metadata note."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    res, cur = {}, {}
    for line in txt.split("\n"):
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip().strip("'")
        if key == "agpr_count" and cur.get("_open"):        # first key of a kernel entry in the YAML
            pass
        cur[key] = val
        if key == "wavefront_size":                          # last key of a kernel entry
            if "symbol" in cur:
                res[cur["symbol"].replace(".kd", "")] = {
                    "vgpr": int(cur.get("vgpr_count", 0)), "agpr": int(cur.get("agpr_count", 0)), "sgpr": int(cur.get("sgpr_count", 0)),
                    "vgpr_spill": int(cur.get("vgpr_spill_count", 0)), "sgpr_spill": int(cur.get("sgpr_spill_count", 0)),
                    "scratch": int(cur.get("private_segment_fixed_size", 0)), "lds": int(cur.get("group_segment_fixed_size", 0)),
                    "wg": int(cur.get("max_flat_workgroup_size", 0))}
            cur = {}
    return res


def histogram(co: str):
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    per, cur = {}, None
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            per[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b", line)
        if m:
            per[cur][m.group(1)] += 1
    return per


def summarise(cnt: collections.Counter):
    valu = {k: v for k, v in cnt.items() if k.startswith("v_") and not k.startswith("v_cmpx")} | {k: v for k, v in cnt.items() if k.startswith("v_cmpx")}
    n_valu = sum(valu.values())
    classes = collections.Counter()
    cyc = arch = 0.0
    for mn, c in valu.items():
        classes[klass(mn)] += c
        cyc += c * cost_of(mn)
        arch += c * arch_cost_of(mn)
    other = {"salu": sum(v for k, v in cnt.items() if k.startswith("s_") and not k.startswith("s_waitcnt") and not k.startswith("s_nop")),
             "waitcnt": sum(v for k, v in cnt.items() if k.startswith("s_waitcnt")),
             "vmem": sum(v for k, v in cnt.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))),
             "lds": sum(v for k, v in cnt.items() if k.startswith("ds_"))}
    return {"valu": n_valu, "classes": dict(classes), "simd_cycles_static": cyc, "avg_cycles_per_valu": cyc / n_valu if n_valu else None,
            "simd_cycles_static_arch": arch, "avg_arch_cycles_per_valu": arch / n_valu if n_valu else None, **other,
            "top": dict(sorted(valu.items(), key=lambda kv: -kv[1])[:12])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(ROOT, "zkir_amd", "libzkir_amd.so"))
    ap.add_argument("--kernels", default="", help="comma-separated substrings of demangled kernel names (default: all)")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    want = [w for w in args.kernels.split(",") if w]
    result = {}
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(args.so, wd):
            meta = notes(co)
            per = histogram(co)
            names = demangle(list(per))
            for sym, cnt in per.items():
                if sym not in meta:                      # device functions that were not inlined, stubs
                    continue
                dm = re.sub(r"^void ", "", names[sym]).replace("(anonymous namespace)::", "")
                dm = re.sub(r"\(.*$", "", dm)
                if want and not any(w in dm for w in want):
                    continue
                result[dm] = {**summarise(cnt), "regs": meta[sym]}
    print(f"# scripts/isa_hist.py over {os.path.relpath(args.so, ROOT)}: STATIC instruction counts per gfx950 kernel (a loop body counts once), VALU by class,")
    print("# cyc/VALU = sum n_i c_i / sum n_i with the measured c_i of profiles/r02_ubench_alu.txt; arch = the same with the architectural costs (4 SIMD-cycles per wave64")
    print("# instruction, 2 for the full-rate 32-bit set: the bench line's ALU bound); registers / spills / scratch from the code object's metadata note")
    hdr = f"{'kernel':58s} {'VALU':>6s} {'mult':>6s} {'add64':>6s} {'copy':>6s} {'copy%':>6s} {'fr32':>6s} {'other':>6s} {'cyc/VALU':>8s} {'arch':>5s} {'SALU':>6s} {'VMEM':>5s} {'LDS':>5s} {'VGPR':>5s} {'AGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch':>7s}"
    print(hdr)
    for name in sorted(result):
        r = result[name]
        c = r["classes"]
        print(f"{name[:58]:58s} {r['valu']:6d} {c.get('multiplier', 0):6d} {c.get('add64', 0):6d} {c.get('copy', 0):6d} {100.0 * c.get('copy', 0) / max(r['valu'], 1):6.1f} "
              f"{c.get('full_rate_32', 0):6d} {c.get('other_valu', 0):6d} {r['avg_cycles_per_valu'] or 0:8.2f} {r['avg_arch_cycles_per_valu'] or 0:5.2f} {r['salu']:6d} {r['vmem']:5d} {r['lds']:5d} "
              f"{r['regs']['vgpr']:5d} {r['regs']['agpr']:5d} {r['regs']['vgpr_spill']:6d} {r['regs']['sgpr_spill']:6d} {r['regs']['scratch']:7d}")
    if args.json:
        sys.path.insert(0, ROOT)
        from zkir_amd.build import sources_sha16
        result["_kernels_sha16"] = sources_sha16()
        json.dump(result, open(args.json, "w"), indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
