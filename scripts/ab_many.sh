#!/bin/bash
# Same-box comparison of several builds (scripts/ubench/bin/libmmx_<tag>.so) on the pair kernels along a minimization
# (nb_states.py), two rounds.   usage: ab_many.sh <workload> <tag> <tag> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
W=$1; shift
for i in 1 2; do for v in "$@"; do echo "== $v"; MMX_LIB=$R/scripts/ubench/bin/libmmx_$v.so python3 scripts/nb_states.py $W 2>&1 | cut -c1-110; done; done
