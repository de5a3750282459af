#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the pair kernel's HBM traffic.
# Usage: bash scripts/profile_gpu.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --cpu-seconds 0 > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/pmc_$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % C, recursive=True)
    if not fs:
        print(C, "no csv"); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    with open("$OUT/pmc_%s_summary.txt" % C, "w") as f:
        for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]:
            line = "%-44s dispatches=%5d  %s per dispatch = %.1f (KB as reported)" % (k, n, C, v / n)
            print(line); f.write(line + "\n")
PY
