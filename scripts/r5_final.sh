#!/bin/bash
# Final evidence of a round on the GPU box: rocprofv3 stats of the driver's command and of the default one, the full GPU suite,
# the launch-path rehearsals with 2 and 4 ranks on the one GPU.  Output under gpurun_out/final_<tag>/.
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
OUT=$R/gpurun_out/final_$TAG; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace20 -- python3 bench.py --steps 20 --warmup 5 > $OUT/bench20_under_rocprof.log 2>&1 || { tail -5 $OUT/bench20_under_rocprof.log; exit 1; }
cp $(find $OUT/trace20 -name "*kernel_stats.csv" | head -1) $OUT/bench20_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace200 -- python3 bench.py > $OUT/bench_default_under_rocprof.log 2>&1 || { tail -5 $OUT/bench_default_under_rocprof.log; exit 1; }
cp $(find $OUT/trace200 -name "*kernel_stats.csv" | head -1) $OUT/bench_default_kernel_stats.csv
rm -rf $OUT/trace20 $OUT/trace200
python3 scripts/kstats.py $OUT/bench20_kernel_stats.csv 8; python3 scripts/kstats.py $OUT/bench_default_kernel_stats.csv 8
echo "profiles done"
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1; tail -3 $OUT/gpu_tests.txt
for n in 2 4; do
  timeout -k 10 600 python3 bench.py --gpus $n --steps 20 --warmup 5 > $OUT/bench_gpus${n}_rehearsal_one_gpu.json 2> $OUT/bench_gpus${n}_rehearsal_one_gpu.err; echo "rehearsal $n: exit $?"
done
python3 scripts/det_hash.py 120 > $OUT/det_hash_final.txt 2>&1; cat $OUT/det_hash_final.txt
