export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/scripts/time_lde.py ${1:-20} > /tmp/kt.log 2>&1
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(sorted(glob.glob("/tmp/kt/*.db"))[0])
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"):
    print(name.replace("void (anonymous namespace)::","").split("(")[0][:60], calls, round(avg,1))
PY
