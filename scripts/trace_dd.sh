#!/bin/bash
# Kernel durations of a decomposed run (loopback communicator, gw_1m on 8 ranks, 40 iterations) under rocprofv3: what the halo's
# own kernels cost per evaluation next to the force kernels.   usage: trace_dd.sh <tag>
TAG=${1:-dd}
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $R
OUT=$R/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/dd_halo_stats.py gw_1m 40 1 8 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
fs = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
ev = max(int(r["Calls"]) for r in rows if "k_history" in r["Name"])
print("evaluations x ranks:", ev)
for r in rows[:24]:
    name = r["Name"].split("(")[0].split("<")[0][-30:]
    print("%-32s calls=%6s avg_us=%8.1f  us per evaluation and rank=%7.2f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / ev))
PY
