#!/bin/bash
# Round-5 quick measurement pass (GPU box): driver's 20-step line x3, default line, chr1_50k, deterministic fingerprints.
# usage: r5_quick.sh <tag>     output: gpurun_out/<tag>/
set -u
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/b20_$i.json 2> $OUT/b20_$i.err || exit 1
done
timeout -k 10 200 python3 bench.py --cpu-seconds 0 > $OUT/bdef.json 2> $OUT/bdef.err || exit 1
timeout -k 10 200 python3 bench.py --cpu-seconds 0 --workload chr1_50k > $OUT/bchr1.json 2> $OUT/bchr1.err || exit 1
timeout -k 10 200 python3 bench.py --cpu-seconds 0 --workload gw_1m > $OUT/b1m.json 2> $OUT/b1m.err || exit 1
timeout -k 10 300 python3 scripts/det_hash.py 120 > $OUT/det_hash.txt 2> $OUT/det_hash.err || exit 1
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/b*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), d["config"].get("workload"), round(d["value"], 1), round(d["ms_per_step"], 4), "ms",
          {k: round(v, 1) for k, v in (d.get("kernel_us_mean") or {}).items() if v})
PY
cat $OUT/det_hash.txt
