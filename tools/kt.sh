#!/bin/bash
# kernel-trace statistics of one flow: bash tools/kt.sh <tag> <variant> [env...]   -> gpurun_out/<tag>_kt_stats.csv (+ top lines on stdout)
TAG=$1; VAR=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
N=${N:-20} rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt --output-format csv -- python3 $R/tools/pmc_run.py $VAR > /dev/null 2>> $OUT/${TAG}_kt.log
find $OUT/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kt_stats.csv \;
cut -d, -f1-4 $OUT/${TAG}_kt_stats.csv | grep -v "at::\|rocprim\|rocclr" | head -${TOP:-14}
