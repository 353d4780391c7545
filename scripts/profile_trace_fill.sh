#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun). Writes under gpurun_out/prof_r01/.
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp
python $R/bench.py --steps 50 --warmup 5 > $OUT/bench_k20.json 2>$OUT/bench_k20.err
python $R/bench.py --steps 20 --warmup 3 --log2-rows 24 --no-cpu-baseline > $OUT/bench_k24.json 2>$OUT/bench_k24.err
rocprofv3 --kernel-trace --stats -d $OUT/kt -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/kt_stdout.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch_stdout.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc_write_stdout.log 2>&1
find $OUT -type f | head -50
ls -la $OUT/kt/* | head
