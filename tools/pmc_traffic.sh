#!/bin/bash
# HBM-traffic counters of every flow: one rocprofv3 --pmc pass per (flow, counter), nothing traced alongside.
# Usage (on the GPU box): bash tools/pmc_traffic.sh <tag> [flows...]   -> gpurun_out/<tag>_pmc/ + gpurun_out/<tag>_pmc_traffic.json
set -e
TAG=${1:-rXX}; shift || true
FLOWS=${@:-dir32 by32 by64 dir64 by16 dir16 enc32 exp32 dec32 exp64 exp16 dec64 sqdir16 sqdir32 sqdir64 sqby16 sqby32 sqby64}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in $FLOWS; do
  var=${f%%[0-9]*}; export BLOCK=${f##*[a-z]}
  export N=3
  OUTJSON=$OUT/$f.json rocprofv3 --pmc FETCH_SIZE -d $OUT/${f}_fetch --output-format csv -- python3 $R/tools/pmc_run.py $var > /dev/null 2>> $OUT/log.txt
  rocprofv3 --pmc WRITE_SIZE -d $OUT/${f}_write --output-format csv -- python3 $R/tools/pmc_run.py $var > /dev/null 2>> $OUT/log.txt
  echo "$f done"
done
cd $R
python3 tools/pmc_traffic.py $OUT > $R/gpurun_out/${TAG}_pmc_traffic.json
