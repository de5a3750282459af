#!/bin/bash
# PMC counters of the pair kernel at a fixed state (run on the GPU box).  Usage: pmc_nb.sh "<counters>" <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $R
OUT=$R/gpurun_out/pmc_$2; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc $1 --output-format csv -d $OUT -- python3 scripts/nb_bench.py gw_200k 0 0 nocensus > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0][-36:]
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in acc:
    if "nb_" in k:
        for c, (v, n) in acc[k].items():
            print("%-36s %-24s per-dispatch %.4g (n=%d)" % (k, c, v / n, n))
PY
