#!/bin/bash
# Profiles of the bench command for profiles/: kernel trace + stats of the default bench.py run, and the
# HBM-traffic counter passes (separate runs per counter, no tracing alongside --pmc).
# Usage (on the GPU box): bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_kt_bench.json 2> $OUT/${TAG}_kt.log
for m in dir by; do
  mode=shrink_directionally; [ $m = by ] && mode=shrink_by
  rocprofv3 --pmc FETCH_SIZE -d $OUT/${TAG}_pmc/pmc_fetch_$m --output-format csv -- python3 $R/bench.py --no-cpu-baseline --mode $mode --steps 4 --warmup 2 > /dev/null 2>> $OUT/${TAG}_kt.log
  rocprofv3 --pmc WRITE_SIZE -d $OUT/${TAG}_pmc/pmc_write_$m --output-format csv -- python3 $R/bench.py --no-cpu-baseline --mode $mode --steps 4 --warmup 2 > /dev/null 2>> $OUT/${TAG}_kt.log
done
cd $R
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc > $OUT/${TAG}_pmc_traffic.json
find $OUT/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
echo done
