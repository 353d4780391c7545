#!/bin/bash
# the LDE alone, several builds of the library side by side on one box: variants_lde.sh <variant> ...   ("default" = the in-tree build)
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}; cd $R
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = default ]; then unset ZKIR_AMD_LIB; else export ZKIR_AMD_LIB=$R/zkir_amd/variants/libzkir_amd_$v.so; fi
  echo "== $v (pass $rep): $(python scripts/exp_lde.py ${K:-20} ${W:-152} 20 2>&1 | grep '^lde')"
done; done
