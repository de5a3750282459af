#!/bin/bash
# Same-box comparison of a WHOLE earlier tree (library + bench.py + Python side) with the working tree, interleaved, three rounds.
# The earlier tree is an export of its commit beside the working tree, built here, so that it travels to the GPU box:
#   mkdir -p scripts/ubench/bin/r3tree && git archive <commit> | tar -x -C scripts/ubench/bin/r3tree
#   (cd scripts/ubench/bin/r3tree && rm -rf profiles tests/golden && python -m multimm_amd.build)
# usage (on the GPU box): ab_r3.sh "<bench flags>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
F=$1
for i in 1 2 3; do for v in r3 now; do
  if [ $v = r3 ]; then D=$R/scripts/ubench/bin/r3tree; else D=$R; fi
  (cd $D && python3 bench.py --cpu-seconds 0 $F 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$F', round(d['value'],1), 'iters/s', round(d['ms_per_step']*1e3,1), 'us/iter', d['evaluations'], {k:round(v,1) for k,v in d['kernel_us_mean'].items()})")
done; done
