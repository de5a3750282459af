#!/bin/bash
# rocprofv3 kernel stats of a python script.  usage: r5_prof_py.sh <tag> <script> [args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $F $OUT/kernel_stats.csv
python3 scripts/kstats.py $OUT/kernel_stats.csv 30
