#!/bin/bash
# Same-box A/B of two builds (scripts/ubench/bin/libmmx_A.so / _B.so) on the half-shell pair kernel at the states a
# minimization passes through (nb_states.py, A B A B): usage ab_states.sh [workload]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
W=${1:-gw_200k}
for i in 1 2; do for v in A B; do echo "== $v"; MMX_LIB=$R/scripts/ubench/bin/libmmx_$v.so python3 scripts/nb_states.py $W 2>&1 | cut -c1-110; done; done
