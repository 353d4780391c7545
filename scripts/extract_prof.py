#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into small text/JSON files under profiles/.

usage: extract_prof.py <prof_dir> <out_prefix> [kernel_substr ...]
  <prof_dir>/kt/*.db          --kernel-trace --stats run      -> <out_prefix>_kernel_stats.txt
  <prof_dir>/pmc_fetch/*.db   --pmc FETCH_SIZE run            \
  <prof_dir>/pmc_write/*.db   --pmc WRITE_SIZE run            -> <out_prefix>_pmc_traffic.json (+ .txt)
HBM traffic follows MI355X_MICROARCH.md §HBM: WRITE_SIZE (KB) is taken as reported (calibrated below against
torch fill kernels of known size in the same run); FETCH_SIZE (KB) is DOUBLED for wide coalesced (16 B/lane)
streaming reads, the gfx950 correction the guide prescribes.
"""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def db(path):
    f = sorted(glob.glob(path + "/*.db"))
    return sqlite3.connect(f[0]) if f else None


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:90]


def main():
    prof, out = sys.argv[1], sys.argv[2]
    keys = sys.argv[3:] or ["trace_fill"]
    lines = []
    c = db(prof + "/kt")
    if c:
        lines.append("# rocprofv3 --kernel-trace --stats : top kernels (name, calls, total_us, avg_us, pct)")
        for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
            lines.append(f"{short(name):90s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}%")
        open(out + "_kernel_stats.txt", "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
    traffic = {}
    cal = []
    for which, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        c = db(prof + "/" + which)
        if not c:
            continue
        q = "select kernel_name, avg(value), min(value), max(value), count(*), avg(duration) from counters_collection where counter_name=? group by kernel_name"
        for name, avg, mn, mx, n, dur in c.execute(q, (counter,)):
            s = short(name)
            if any(k in s for k in keys):
                traffic.setdefault(s, {})[counter + "_KB_avg"] = avg
                traffic[s][counter + "_launches"] = n
                traffic[s][counter + "_avg_ns"] = dur
            if "FillFunctor" in s:
                cal.append(f"{counter} calibration: {s} reported {avg} KB")
    for s, t in traffic.items():
        # gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 for wide coalesced reads.  The strided NTT passes read 64-B
        # runs (16 consecutive words per tile row, C = 4): their requests are counted in full (the raw value equals the
        # algorithmic 4 B/element exactly), so no correction there.
        factor = 2                       # B8 layout: every kernel reads >= 128-byte runs with 16-byte lanes (the strided NTT tiles are 128-byte rows)
        t["fetch_correction_factor"] = factor
        fetch = t.get("FETCH_SIZE_KB_avg", 0.0) * 1024 * factor
        write = t.get("WRITE_SIZE_KB_avg", 0.0) * 1024
        t["hbm_read_bytes_corrected"] = fetch
        t["hbm_write_bytes"] = write
        t["hbm_bytes_per_launch"] = fetch + write
    if traffic:
        from zkir_amd.build import sources_sha16
        traffic["_kernels_sha16"] = sources_sha16()
        json.dump(traffic, open(out + "_pmc_traffic.json", "w"), indent=1, sort_keys=True)
        txt = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per launch"] + cal
        for s, t in traffic.items():
            if s.startswith("_"):
                txt.append(f"# kernels_sha16: {t}")
                continue
            txt.append(s)
            for k in sorted(t):
                txt.append(f"    {k:32s} {t[k]}")
        open(out + "_pmc_traffic.txt", "w").write("\n".join(txt) + "\n")
        print("\n".join(txt))


if __name__ == "__main__":
    main()
