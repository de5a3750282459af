#!/bin/bash
# kernel_choice.py (half shell forced / full shell forced, minimization from the lattice) with two builds (libmmx_A / _B), at several sizes.
# usage: ab_choice.sh <n_beads> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for nb in "$@"; do for v in A B; do echo "== $v $nb"; MMX_LIB=$R/scripts/ubench/bin/libmmx_$v.so python3 scripts/kernel_choice.py gw_200k 200,2000 $nb 2>&1 | grep forced; done; done
