#!/bin/bash
# per-kernel split of the LDE alone for a list of "k W" configurations: exp_lde.sh "20 152" "20 2432" "24 152"
export TMPDIR=/tmp; cd /tmp
for cfg in "$@"; do
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/scripts/exp_lde.py $cfg > /tmp/kt.log 2>&1
  grep "^lde" /tmp/kt.log
  python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(sorted(glob.glob("/tmp/kt/*.db"))[0])
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 6"):
    if "ntt" in name or "lde" in name: print("   ", name.replace("void (anonymous namespace)::","").split("(")[0][:60], calls, round(avg,1))
PY
done
