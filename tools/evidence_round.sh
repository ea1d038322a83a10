#!/bin/bash
# Everything profiles/ holds for a round, in one call on the GPU box (about 10 minutes):
#   bash tools/evidence_round.sh r03_v2
# kernel trace + stats of the default bench command, HBM-traffic counters of every flow, SQ counters of the kernels the
# bench line quotes, the bench line itself without a profiler, the CPU-baseline table.
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
bash $R/tools/profile_round.sh $TAG > $OUT/${TAG}_profile.log 2>&1
echo "profile_round done"
bash $R/tools/sq_counters.sh ${TAG}_dir32 dir > /dev/null 2>&1
BLOCK=32 bash $R/tools/sq_counters.sh ${TAG}_by32 by > /dev/null 2>&1
BLOCK=64 bash $R/tools/sq_counters.sh ${TAG}_by64 by > /dev/null 2>&1
bash $R/tools/sq_counters.sh ${TAG}_enc enc > /dev/null 2>&1
FILTER=4 bash $R/tools/sq_counters.sh ${TAG}_exp exp > /dev/null 2>&1
bash $R/tools/sq_counters.sh ${TAG}_dec dec > /dev/null 2>&1
cd $R
python3 tools/sq_summary.py $OUT/${TAG}_dir32_sq.txt $OUT/${TAG}_by32_sq.txt $OUT/${TAG}_by64_sq.txt $OUT/${TAG}_enc_sq.txt $OUT/${TAG}_exp_sq.txt $OUT/${TAG}_dec_sq.txt > $OUT/${TAG}_sq_summary.json
echo "sq done"
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench done"
python3 tools/cpu_baseline_table.py > $OUT/${TAG}_cpu_baseline.json 2> $OUT/${TAG}_cpu_baseline.err
echo "cpu table done"
