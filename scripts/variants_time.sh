#!/bin/bash
# Time the commit stages with several builds of the library on ONE box (box-to-box spread is ~3 %): scripts/variants_time.sh <tag> <variant> ...
# (variants: zkir_amd/variants/libzkir_amd_<variant>.so, made by zkir_amd.build.build_variant; "default" = the in-tree build;
#  W=<columns> in the environment overrides the matrix width the script commits, for builds with another main-trace width)
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}; TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = default ]; then unset ZKIR_AMD_LIB; else export ZKIR_AMD_LIB=$R/zkir_amd/variants/libzkir_amd_$v.so; fi
  echo "== $v (pass $rep)"; (cd $R && timeout 300 python scripts/time_commit.py ${K:-20} 2>&1 | tail -6)
done; done | tee $OUT/variants_time.txt
