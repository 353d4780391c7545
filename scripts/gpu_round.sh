#!/bin/bash
# One GPU-box session (run through gpurun): GPU parity tests, the bench line, and rocprofv3 kernel-trace summaries for the three
# single-GPU BASELINE configs.  Usage: gpu_round.sh <tag> [tests|bench|prof ...]   (default: all three)
# Counter (PMC) passes are separate runs with --kernel-trace only (never combined with other trace domains).
export TMPDIR=/tmp
TAG=${1:-r04}; shift
WHAT=${@:-tests bench prof}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
summ() {  # <db dir> <out file>: top kernels of a rocprofv3 --kernel-trace --stats run
python - "$1" "$2" <<'PY'
import sqlite3, glob, sys
dbs = sorted(glob.glob(sys.argv[1] + "/*.db"))
if not dbs: sys.exit("no rocpd db under " + sys.argv[1])
c = sqlite3.connect(dbs[0])
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 40").fetchall()
with open(sys.argv[2], "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats: kernel, calls, total us, average us, share\n")
    for name, calls, total, avg, pct in rows:
        s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:80]
        f.write(f"{s:80s} {calls:6d} {total:12.1f} us {avg:10.2f} us {pct:6.2f}%\n")
print(open(sys.argv[2]).read())
PY
}
for w in $WHAT; do case $w in
tests) (cd $R && timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log;;
bench) (cd $R && timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err); head -c 3000 $OUT/bench.json; echo; tail -3 $OUT/bench.err;;
prof)
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_c1 -o b -- python $R/bench.py --no-cpu-baseline --no-prove --no-by-config > $OUT/kt_c1.log 2>&1
  summ $OUT/kt_c1 $OUT/${TAG}_config1_kernel_stats.txt
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_c2 -o b -- python $R/bench.py --only-config 2 > $OUT/kt_c2.log 2>&1
  summ $OUT/kt_c2 $OUT/${TAG}_config2_kernel_stats.txt
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_c4 -o b -- python $R/bench.py --only-config 4 > $OUT/kt_c4.log 2>&1
  summ $OUT/kt_c4 $OUT/${TAG}_config4_kernel_stats.txt
  rm -rf $OUT/kt_c1 $OUT/kt_c2 $OUT/kt_c4;;
pmc)
  export ZKIR_EXEC_STREAM=0   # whole-run K1 launches only: the streaming zkir_exec adds partial-range launches to the per-kernel averages
  for c in FETCH_SIZE WRITE_SIZE; do d=$(echo $c | tr A-Z a-z | sed 's/_size//'); 
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$d -o b -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-prove --no-by-config > $OUT/pmc_$d.log 2>&1; done
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_valu -o b -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-prove --no-by-config > $OUT/pmc_valu.log 2>&1
  python $R/scripts/extract_prof.py $OUT $OUT/${TAG}_bench_commit trace_fill main_trace lde_middle ntt_strided leaf_hash compress subtree | cut -c1-150
  python $R/scripts/extract_valu.py $OUT/pmc_valu $OUT/${TAG}_bench_commit_valu_busy.txt
  rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_valu;;
pmc4)   # HBM traffic of the configs[4] witness kernels (SHA chain, 2^22 cycles): separate counter passes, --kernel-trace only
  export ZKIR_EXEC_STREAM=0
  for c in FETCH_SIZE WRITE_SIZE; do d=$(echo $c | tr A-Z a-z | sed 's/_size//');
    timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$d -o b -- python $R/bench.py --only-config 4 > $OUT/pmc4_$d.log 2>&1; done
  python $R/scripts/extract_prof.py $OUT $OUT/${TAG}_config4 trace_fill memops_expand memops_sort memops_row_offsets memops_segment sha256_chip | cut -c1-150
  rm -rf $OUT/pmc_fetch $OUT/pmc_write;;
esac; done
