#!/bin/bash
# rocprofv3 kernel stats of the driver's 20-step command (+ optional engine options).  usage: r5_prof.sh <tag> [bench args...]
set -u
TAG=${1:-r5p}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $F $OUT/kernel_stats.csv
python3 scripts/kstats.py $OUT/kernel_stats.csv 14
grep "^{" $OUT/trace.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['ms_per_step'])"
