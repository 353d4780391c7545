export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/m3; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o c -- python $R/scripts/time_prove_modes.py 20 > $OUT/stdout.log 2>&1
python - <<PY
import sqlite3, glob
c = sqlite3.connect(sorted(glob.glob("$OUT/kt/*.db"))[0])
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 45").fetchall()
for name, calls, total, avg, pct in rows:
    s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    print(f"{s:70s} {calls:6d} {total:12.1f} us {avg:10.2f} us {pct:6.2f}%")
PY
