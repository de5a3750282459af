#!/bin/bash
# Builds libmmx from a source tree into scripts/ubench/bin/libmmx_<tag>.so for same-box A/B runs (MMX_LIB=...).
#   scripts/build_variant.sh <tag> [git-rev]     (no rev: the working tree; with rev: a clean checkout of it under /tmp)
set -e
cd "$(dirname "$0")/.."
TAG=$1; REV=${2:-}
mkdir -p scripts/ubench/bin
SRC=multimm_amd/csrc/mmx_api.hip
if [ -n "$REV" ]; then
  rm -rf /tmp/mmx_wt_$TAG; git worktree prune; git worktree add -f --detach /tmp/mmx_wt_$TAG $REV >/dev/null 2>&1
  SRC=/tmp/mmx_wt_$TAG/multimm_amd/csrc/mmx_api.hip
fi
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-slp-vectorize -o scripts/ubench/bin/libmmx_$TAG.so $SRC
[ -n "$REV" ] && git worktree remove --force /tmp/mmx_wt_$TAG
ls -la scripts/ubench/bin/libmmx_$TAG.so
