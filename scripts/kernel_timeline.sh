#!/bin/bash
# Kernel TIMELINE of the last proof of scripts/time_prove.py (rocprofv3 --kernel-trace): start offset, duration and the gap before each dispatch — where a latency-bound stage
# (FRI, the tree tails) spends its time.  Usage (through gpurun): scripts/kernel_timeline.sh <tag> [log2_rows=20]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-timeline}; mkdir -p $OUT; cd /tmp
rm -rf $OUT/kt; rocprofv3 --kernel-trace -d $OUT/kt -o t -- python $R/scripts/time_prove.py ${2:-20} > $OUT/time_prove.log 2>&1
python - "$OUT/kt" "$OUT/${1:-timeline}_prove_timeline.txt" <<'PY'
import sqlite3, glob, sys
c = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/*.db"))[0])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
rows = c.execute(f"select name, start, end from {view} order by start").fetchall()
# the last proof: from the last main_trace_kernel on
idx = max(i for i, r in enumerate(rows) if "main_trace_kernel" in r[0])
rows = rows[idx:]
t0 = rows[0][1]
out = ["# kernel timeline of one proof (rocprofv3 --kernel-trace): start offset us, duration us, gap since the previous kernel's end us, name"]
prev = None
for name, st, en in rows:
    s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    gap = (st - prev) / 1e3 if prev is not None else 0.0
    out.append(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:9.1f} {gap:8.1f}  {s}")
    prev = en
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $OUT/kt
