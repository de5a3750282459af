#!/bin/bash
# One pass over every number DESIGN.md quotes (run on the GPU box): headline bench with the CPU baseline, size sweep,
# MD rates, time to solution, decomposed run through the loopback communicator.  Output: gpurun_out/measure_<tag>/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/measure_$TAG; mkdir -p $OUT
timeout -k 10 500 python3 bench.py > $OUT/bench_gw200k.json 2> $OUT/bench_gw200k.err && echo "headline done"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_gw200k_20steps.json 2> $OUT/bench_gw200k_20steps.err && echo "20-step line done"
for w in region_5k chr1_50k gw_1m; do
  timeout -k 10 300 python3 bench.py --workload $w --cpu-seconds 0 --no-dd-leg > $OUT/bench_$w.json 2>/dev/null && echo "$w done"
done
timeout -k 10 200 python3 bench.py --cpu-seconds 0 --no-dd-leg --cutoff 0 --workload chr1_50k --steps 50 > $OUT/bench_nocutoff_50k.json 2>/dev/null && echo "nocutoff done"
for w in region_5k chr1_50k gw_200k; do
  timeout -k 10 200 python3 scripts/md_bench.py $w 2000 200 | tail -1 >> $OUT/md.txt
done
echo "md done"
for w in chr1_50k gw_200k gw_1m; do
  timeout -k 10 300 python3 scripts/converge.py $w | tail -2 >> $OUT/converge.txt
done
echo "converge done"
timeout -k 10 600 python3 scripts/dd_projection.py gw_1m 150 1,2,4,8 > $OUT/dd_projection.txt 2>&1
timeout -k 10 400 python3 scripts/dd_halo_stats.py gw_1m 150 1500 8,4,2 > $OUT/dd_halo_stats.txt 2>&1
echo "dd done"
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print(os.path.basename(f), d["config"].get("n_beads"), round(d["value"], 1), d["unit"], round(d["ms_per_step"], 4), "ms",
          {k: round(v, 1) for k, v in d.get("kernel_us_mean", {}).items()}, "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
cat $OUT/md.txt $OUT/converge.txt; grep "^world\|beads, relaxed" $OUT/dd_projection.txt; cat $OUT/dd_halo_stats.txt
