#!/bin/bash
# Prover kernels (scripts/time_prove.py = zkir_prove on a 2^k-cycle fib trace): rocprofv3 kernel stats, then counter passes in their OWN runs
# (--kernel-trace only, never combined with other trace domains): VALU issue (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE)
# and HBM traffic (FETCH_SIZE, WRITE_SIZE).  Usage: pmc_prove.sh <tag> [k=20]  ->  gpurun_out/<tag>/<tag>_prove_{kernel_stats,valu_busy,pmc_traffic}_2p<k>.*
export TMPDIR=/tmp
TAG=${1:-r04}; K=${2:-20}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_p -o c -- python $R/scripts/time_prove.py $K > $OUT/prove_kt_2p$K.log 2>&1
python - "$OUT/kt_p" "$OUT/${TAG}_prove_kernel_stats_2p$K.txt" <<'PY'
import sqlite3, glob, sys
c = sqlite3.connect(sorted(glob.glob(sys.argv[1] + "/*.db"))[0])
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 40").fetchall()
with open(sys.argv[2], "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats of scripts/time_prove.py (3 proofs + setup): kernel, calls, total us, average us, share\n")
    for name, calls, total, avg, pct in rows:
        s = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        f.write(f"{s:70s} {calls:6d} {total:12.1f} us {avg:10.2f} us {pct:6.2f}%\n")
print(open(sys.argv[2]).read())
PY
tail -12 $OUT/prove_kt_2p$K.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_valu_p -o c -- python $R/scripts/time_prove.py $K > $OUT/prove_pmc_valu.log 2>&1
python $R/scripts/extract_valu.py $OUT/pmc_valu_p $OUT/${TAG}_prove_valu_busy_2p$K.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write
for c in FETCH_SIZE WRITE_SIZE; do d=$(echo $c | tr A-Z a-z | sed 's/_size//');
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$d -o c -- python $R/scripts/time_prove.py $K > $OUT/prove_pmc_$d.log 2>&1; done
python $R/scripts/extract_prof.py $OUT $OUT/${TAG}_prove_2p$K quotient_kernel deep_kernel bary_dot bary_weights leaf_hash lde_middle ntt_strided main_trace aux_rows lookup_index fri_fold fri_leaf | cut -c1-160
rm -rf $OUT/kt_p $OUT/pmc_valu_p $OUT/pmc_fetch $OUT/pmc_write
