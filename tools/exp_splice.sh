R=${GRAFT_REPO_ROOT:-$(pwd)}
for e in 0 4 5; do
  if [ $e = 0 ]; then unset PXZ_LIB; else export PXZ_LIB=$R/pixlzr-rust_amd/csrc/libpixlzr_hip_exp$e.so; fi
  N=20 TOP=12 bash tools/kt.sh spl$e enc | grep "splice" | cut -d, -f1,4 | sed "s/^/exp $e: /"
done
