set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/sqv; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "lod 3 0.25 dir_lod" "c1x1 3 0.25 dir_full" "c4x4 3 1 dir_full" "c8x8 3 2 dir_full" "c16 3 4 dir_full" "clone 3 16 dir_full" "mix 0 16 dir_full"; do
  set -- $cfg
  DIST=$2 FACTOR=$3 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/$1 --output-format csv -- python3 $R/tools/pmc_run.py $4 > /dev/null 2>> $OUT/log.txt
done
cd $R
for n in lod c1x1 c4x4 c8x8 c16 clone mix; do echo "== $n"; python3 tools/pmc_summary.py $OUT/$n | grep -i "shrink32\|lod" | cut -c1-220; done
