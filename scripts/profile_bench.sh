#!/bin/bash
# rocprofv3 evidence for `python bench.py` (run on the GPU box via gpurun). Usage: profile_bench.sh <round-tag>
# The profiled command is `bench.py --no-cpu-baseline --no-prove`: the timed steps only (the proof that bench.py runs afterwards
# launches the same kernels at other sizes and would blur the per-kernel averages).  Counter passes are separate runs with --kernel-trace only (never combined with other trace domains).
export TMPDIR=/tmp
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp
python $R/bench.py > $OUT/bench.json 2>$OUT/bench.err
python $R/bench.py --stage trace_fill --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_trace_fill.json 2>>$OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o b -- python $R/bench.py --no-cpu-baseline --no-prove > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o b -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-prove > $OUT/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o b -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-prove > $OUT/pw.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_valu -o b -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-prove > $OUT/pv.log 2>&1
python $R/scripts/extract_prof.py $OUT $OUT/summary trace_fill main_trace lde_middle ntt_strided leaf_hash compress subtree | cut -c1-150
python $R/scripts/extract_valu.py $OUT/pmc_valu $OUT/summary_valu_busy.txt
head -c 1500 $OUT/bench.json; echo; tail -3 $OUT/bench.err
