#!/bin/bash
# Round-5 check after a kernel change (GPU box): deterministic fingerprints against profiles/r05/det_hash_before.txt, the GPU tests
# given as arguments (default: parity + faults), the driver's 20-step line.    usage: r5_check.sh <tag> [pytest args...]
set -u
TAG=${1:-r5c}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 300 python3 scripts/det_hash.py 120 > $OUT/det_hash.txt 2> $OUT/det_hash.err || { tail -5 $OUT/det_hash.err; exit 1; }
if diff -q $OUT/det_hash.txt profiles/r05/det_hash_before.txt > /dev/null; then echo "deterministic fingerprints: identical"; else echo "deterministic fingerprints DIFFER"; diff $OUT/det_hash.txt profiles/r05/det_hash_before.txt; fi
ARGS="$@"; [ -z "$ARGS" ] && ARGS="tests/test_gpu_parity.py tests/test_gpu_faults.py"
timeout -k 10 900 python3 -m pytest $ARGS -x -q -m gpu > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.txt
for i in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/b20_$i.json 2> $OUT/b20_$i.err || exit 1
done
timeout -k 10 200 python3 bench.py --cpu-seconds 0 > $OUT/bdef.json 2> $OUT/bdef.err || exit 1
timeout -k 10 200 python3 bench.py --cpu-seconds 0 --workload chr1_50k > $OUT/bchr1.json 2> $OUT/bchr1.err || exit 1
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/b*.json")):
    d = json.load(open(f))
    print(os.path.basename(f), d["config"].get("workload","")[:8], round(d["value"], 1), round(d["ms_per_step"], 4), "ms",
          {k: round(v, 1) for k, v in (d.get("kernel_us_mean") or {}).items() if v})
PY
