#!/bin/bash
# Same-box A/B of two builds of libmmx.so on the pair kernel (run on the GPU box):
#   scripts/ubench/bin/libmmx_A.so and libmmx_B.so, loaded through MMX_LIB; prints lattice / relaxed times twice.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2; do for v in A B; do echo $v; MMX_LIB=$R/scripts/ubench/bin/libmmx_$v.so python3 scripts/nb_bench.py gw_200k 300 0 nocensus 2>&1 | grep nb_variant; done; done
