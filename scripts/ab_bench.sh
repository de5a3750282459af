#!/bin/bash
# Same-box A/B of two builds (scripts/ubench/bin/libmmx_A.so / _B.so) on whole minimizations: bench.py lines A B A B.
#   usage: ab_bench.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2 3; do for v in A B; do MMX_LIB=$R/scripts/ubench/bin/libmmx_$v.so python3 bench.py --cpu-seconds 0 "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.load(sys.stdin)
print('$v', round(d['value'], 1), 'it/s', round(d['ms_per_step'], 4), 'ms', {k: round(v, 1) for k, v in d['kernel_us_mean'].items()})"; done; done
