#!/bin/bash
# Everything profiles/ holds for a round, on the GPU box, in THREE calls (gpurun limits a call to 20 minutes):
#   bash tools/evidence_round.sh r04_v1 a    kernel trace + stats of the default bench command, HBM-traffic counters of every flow
#   bash tools/evidence_round.sh r04_v1 b    SQ counters of the kernels the bench line quotes (8 x 8K, 32x32 and 64x64 shrink_by)
#   bash tools/evidence_round.sh r04_v1 c    SQ counters of the other tile sizes and the decode side, the bench line itself without
#                                            a profiler, the CPU-baseline table
set -e
TAG=${1:-rXX}
PART=${2:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
if [ "$PART" = a ]; then
  bash $R/tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1
  echo "profile_round done"
elif [ "$PART" = b ]; then
  bash $R/tools/sq_counters.sh ${TAG}_dir32 dir > /dev/null 2>&1
  BLOCK=32 bash $R/tools/sq_counters.sh ${TAG}_by32 by > /dev/null 2>&1
  BLOCK=64 bash $R/tools/sq_counters.sh ${TAG}_by64 by > /dev/null 2>&1
  bash $R/tools/sq_counters.sh ${TAG}_enc enc > /dev/null 2>&1
  FILTER=4 bash $R/tools/sq_counters.sh ${TAG}_exp exp > /dev/null 2>&1
  bash $R/tools/sq_counters.sh ${TAG}_dec dec > /dev/null 2>&1
  echo "sq (b) done"
else
  BLOCK=16 bash $R/tools/sq_counters.sh ${TAG}_dir16 dir > /dev/null 2>&1
  BLOCK=64 bash $R/tools/sq_counters.sh ${TAG}_dir64 dir > /dev/null 2>&1
  FILTER=4 BLOCK=16 bash $R/tools/sq_counters.sh ${TAG}_exp16 exp > /dev/null 2>&1
  FILTER=4 BLOCK=64 bash $R/tools/sq_counters.sh ${TAG}_exp64 exp > /dev/null 2>&1
  python3 tools/sq_summary.py $OUT/${TAG}_*_sq.txt > $OUT/${TAG}_sq_summary.json  # (of THIS call only: gpurun_out does not travel -- run it again at home over all parts)
  echo "sq (c) done"
  python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  echo "bench done"
  python3 tools/cpu_baseline_table.py > $OUT/${TAG}_cpu_baseline.json 2> $OUT/${TAG}_cpu_baseline.err
  echo "cpu table done"
fi
