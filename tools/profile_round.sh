#!/bin/bash
# Profiles of the bench command for profiles/: kernel trace + stats of the default bench.py run, then the HBM-traffic
# counter passes of every flow (tools/pmc_traffic.sh: separate runs per counter, no tracing alongside --pmc).
# Usage (on the GPU box): bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_kt_bench.json 2> $OUT/${TAG}_kt.log
cd $R
find $OUT/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
bash tools/pmc_traffic.sh $TAG
echo done
