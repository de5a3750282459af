#!/bin/bash
# FETCH_SIZE of two pair-kernel variants at the same state (run on the GPU box).  Usage: pmc_nb_ab.sh <variantA,variantB> <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $R
OUT=$R/gpurun_out/pmc_$2; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 scripts/nb_bench.py gw_200k 0 $1 nocensus > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(fs[0])):
    if "nb_clusters" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in acc.items():
    print("%-40s FETCH_SIZE per dispatch %.4g (n=%d)" % (k, v / n, n))
PY
