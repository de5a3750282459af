#!/bin/bash
# Per-kernel bandwidth evidence for the per-bead kernels (run on the GPU box):  rocprofv3 --kernel-trace --stats of a
# minimization with the bonded terms as SEPARATE kernels (fused_bonded = 0), plus FETCH_SIZE / WRITE_SIZE in their own
# passes, at a size where the kernels are bandwidth-bound.   usage: profile_kernels.sh <tag> <workload> <n_beads> <steps>
set -u
TAG=${1:-r02}; WL=${2:-gw_1m}; NB=${3:-0}; STEPS=${4:-60}
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $R
OUT=$R/gpurun_out/kern_${TAG}_${WL}_${NB}; mkdir -p $OUT
ARGS="--workload $WL --n-beads $NB --steps $STEPS --warmup 5 --cpu-seconds 0 --separate-bonded --profile-every 0"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py $ARGS > $OUT/pmc_$C.log 2>&1
done
python3 scripts/kernel_bandwidth.py $OUT $WL $NB > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
