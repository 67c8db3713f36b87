for lib in hip maha_noepi maha_ring0 maha_rt4; do
  echo "== $lib"
  RUNIA_LIB=$GRAFT_REPO_ROOT/runia_core_amd/librunia_$lib.so python tools/ablate/run_maha.py 262144 2048 10
done
