#!/bin/bash
# Arbitrary counters for the prover kernels: one rocprofv3 run per counter GROUP (groups separated by '|'; --kernel-trace only), averages per kernel.
# usage: pmc_counters.sh <tag> "<C1 C2 ..|C3 C4 ..>" [k=20] [kernel substrings..]   -> gpurun_out/<tag>/counters_2p<k>.txt
export TMPDIR=/tmp
TAG=$1; GROUPS_=$2; K=${3:-20}; shift 3
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
: > $OUT/counters_2p$K.txt
IFS='|' read -ra GR <<< "$GROUPS_"
i=0
for g in "${GR[@]}"; do i=$((i+1)); rm -rf $OUT/pmc_g$i
  timeout 600 rocprofv3 --kernel-trace --pmc $g -d $OUT/pmc_g$i -o c -- python $R/scripts/time_prove.py $K > $OUT/pmc_g$i.log 2>&1
  python - "$OUT/pmc_g$i" "$@" >> $OUT/counters_2p$K.txt <<'PY'
import sqlite3, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*.db"))
keys = sys.argv[2:] or ["quotient", "deep_kernel", "bary_dot"]
if f:
    c = sqlite3.connect(f[0])
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    for name, ctr, v, n, dur in sorted(rows):
        s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
        if any(k in s for k in keys): print(f"{s:50s} {ctr:28s} {v:16.0f}  launches {n:3d}  avg_us {dur / 1e3:9.1f}")
PY
  rm -rf $OUT/pmc_g$i
done
cat $OUT/counters_2p$K.txt
