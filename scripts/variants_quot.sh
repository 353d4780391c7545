R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for v in default qw3 qw5 qw6; do
  if [ "$v" = default ]; then unset ZKIR_AMD_LIB; else export ZKIR_AMD_LIB=$R/zkir_amd/variants/libzkir_amd_$v.so; fi
  echo "== $v (pass $rep)"; timeout 300 python scripts/time_prove.py 20 2>&1 | grep -E "quotient|prove wall"
done; done
