#!/bin/bash
# SQ counters of one kernel variant (tools/pmc_run.py <variant>), one rocprofv3 --pmc pass per counter group, summarised
# per kernel.  Usage (on the GPU box): bash tools/sq_counters.sh <tag> <variant>   -> gpurun_out/<tag>_sq.txt
set -e
TAG=${1:-rXX}; VAR=${2:-dir_full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/${TAG}_sq/g$i --output-format csv -- python3 $R/tools/pmc_run.py $VAR > /dev/null 2>> $OUT/${TAG}_sq.log || echo "group $i failed: $grp" >> $OUT/${TAG}_sq.log
done
cd $R
python3 tools/pmc_summary.py $OUT/${TAG}_sq/g* > $OUT/${TAG}_sq.txt
cat $OUT/${TAG}_sq.txt
