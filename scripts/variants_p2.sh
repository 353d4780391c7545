#!/bin/bash
# Poseidon2 / quotient kernel variants on ONE box: per-kernel average durations (rocprofv3 --kernel-trace --stats over scripts/time_commit.py / time_prove.py) and, in a
# separate counter pass, SQ_INSTS_VALU per launch.  Usage (through gpurun): scripts/variants_p2.sh <tag> <variant> ...   ("default" = the in-tree build)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
kstats() {  # <db dir> <kernel substrings...>
python - "$@" <<'PY'
import sqlite3, glob, sys
c = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/*.db"))[0])
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 60").fetchall():
    s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    if any(k in s for k in sys.argv[2:]): print(f"   {s:60s} {calls:5d} calls {avg:10.2f} us avg")
PY
}
pmc() {  # <db dir> <kernel substrings...>
python - "$@" <<'PY'
import sqlite3, glob, sys
c = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/*.db"))[0])
for name, ctr, v, n in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall():
    s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    if any(k in s for k in sys.argv[2:]): print(f"   {s:60s} {ctr:18s} {v:16.0f}  ({n} launches)")
PY
}
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = default ]; then unset ZKIR_AMD_LIB; else export ZKIR_AMD_LIB=$R/zkir_amd/variants/libzkir_amd_$v.so; fi
  echo "== $v (pass $rep)"
  rm -rf $OUT/kt; rocprofv3 --kernel-trace --stats -d $OUT/kt -o c -- python $R/scripts/time_commit.py 20 > $OUT/log_commit_$v.txt 2>&1; grep -E "merkle|total" $OUT/log_commit_$v.txt
  kstats $OUT/kt leaf_hash compress_kernel subtree
  rm -rf $OUT/kt; rocprofv3 --kernel-trace --stats -d $OUT/kt -o c -- python $R/scripts/time_prove.py 20 > $OUT/log_prove_$v.txt 2>&1; grep -E "quotient|openings|fri|prove wall" $OUT/log_prove_$v.txt
  kstats $OUT/kt quotient_kernel bary_dot subtree
  rm -rf $OUT/kt
  if [ $rep = 1 ]; then
    rm -rf $OUT/pmc; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -d $OUT/pmc -o c -- python $R/scripts/time_commit.py 20 > /dev/null 2>&1
    pmc $OUT/pmc leaf_hash compress_kernel; rm -rf $OUT/pmc
  fi
done; done 2>&1 | tee $OUT/${TAG}_variants.txt
