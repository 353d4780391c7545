# quotient / bary / deep kernels at different occupancy targets (variants built with zkir_amd.build.build_variant): per-kernel averages of scripts/time_prove.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-variants}; mkdir -p $OUT; cd /tmp
for rep in 1 2; do for v in default qw4 qw2; do
  if [ "$v" = default ]; then unset ZKIR_AMD_LIB; else export ZKIR_AMD_LIB=$R/zkir_amd/variants/libzkir_amd_$v.so; fi
  rm -rf $OUT/kt_$v; rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o c -- python $R/scripts/time_prove.py ${2:-20} > $OUT/log_$v.txt 2>&1
  echo "== $v (pass $rep)"; grep -E "quotient|openings|deep|prove wall" $OUT/log_$v.txt
  python - "$OUT/kt_$v" <<'PY'
import sqlite3, glob, sys
c = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/*.db"))[0])
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 40").fetchall():
    s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    if any(k in s for k in ("quotient", "bary_dot", "deep_kernel", "leaf_hash", "lde_middle")): print(f"   {s:60s} {calls:5d} {avg:10.2f} us")
PY
  rm -rf $OUT/kt_$v
done; done
