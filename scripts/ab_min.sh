#!/bin/bash
# Same-box A/B of two builds of libmmx.so on the whole minimizer iteration (run on the GPU box):
#   scripts/ubench/bin/libmmx_A.so and libmmx_B.so through MMX_LIB; headline workload, twice each, interleaved.
#   usage: ab_min.sh "<bench flags for A>" "<bench flags for B>"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
FA=$1; FB=$2
for i in 1 2; do for v in A B; do
  if [ $v = A ]; then F=$FA; else F=$FB; fi
  MMX_LIB=$R/scripts/ubench/bin/libmmx_$v.so python3 bench.py --cpu-seconds 0 $F 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$F', round(d['value'],1), 'iters/s', round(d['ms_per_step']*1e3,1), 'us/iter', {k:round(v,1) for k,v in d['kernel_us_mean'].items()})"
done; done
