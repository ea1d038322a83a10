#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by up to 10 %): kernel-trace averages of a flow for each, twice.
#   bash tools/ab.sh <variant of tools/pmc_run.py> <lib A> <lib B> [substring of the kernel names]     (env of pmc_run.py applies)
VAR=$1; A=$2; B=$3; PAT=${4:-pxz::}
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in $A $B $A $B; do
  export PXZ_LIB=$R/$lib
  tag=ab_$(basename $lib .so)
  N=${N:-20} bash tools/kt.sh $tag $VAR > /dev/null
  python3 - "$R/gpurun_out/${tag}_kt_stats.csv" "$PAT" "$(basename $lib .so)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"] and "at::" not in r["Name"]:
        print("%-22s %-60s calls %4s  avg %9.1f us" % (sys.argv[3], r["Name"].replace("void pxz::", "").replace("pxz::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
