#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_commit; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o c -- python $R/scripts/time_commit.py ${1:-20} > $OUT/stdout.log 2>&1
python $R/scripts/extract_prof.py $OUT $OUT/summary none | cut -c1-150
