#!/bin/bash
# Kernel timeline of the default bench (run on the GPU box): per-kernel mean duration and the mean gap to the next kernel of
# the stream, over the timed iterations.   usage: trace_gaps.sh <tag> [bench args]
TAG=${1:-t}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $R
OUT=$R/gpurun_out/trace_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --cpu-seconds 0 "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]          # skip warm-up / setup
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
for a, b in zip(rows, rows[1:]):
    k = a["Kernel_Name"].split("(")[0].split("<")[0][-28:]
    dur[k] += int(a["End_Timestamp"]) - int(a["Start_Timestamp"]); cnt[k] += 1
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 200000: gap[k] += g
ev = max(cnt.get("k_history", 1), 1)
tot = 0.0
for k in sorted(dur, key=lambda k: -dur[k]):
    print("%-30s n=%5d  mean %8.2f us  gap after %6.2f us  per evaluation %8.2f us" % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3, (dur[k] + gap[k]) / ev / 1e3))
    tot += (dur[k] + gap[k]) / ev / 1e3
print("sum per evaluation %.1f us" % tot)
PY
