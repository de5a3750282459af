#!/bin/bash
# Register / LDS / scratch usage of every kernel of libmmx (hipcc -Rpass-analysis=kernel-resource-usage); argument:
# a regex on the demangled kernel name (default: the pair kernels).   usage: scripts/kernel_resources.sh [regex]
pat=${1:-'k_nb_n3|k_nb_clusters_j'}
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage \
  -o /tmp/libmmx_res.so multimm_amd/csrc/mmx_api.hip 2>&1 | c++filt | python3 -c "
import re, sys
pat = re.compile(sys.argv[1])
name = None
row = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (.*)', line)
    if m:
        if name and pat.search(name): print(name[:110], row)
        name, row = m.group(1).strip(), {}
        continue
    m = re.search(r'remark:\s+(VGPRs|TotalSGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m: row[m.group(1).split(' [')[0]] = int(m.group(2))
if name and pat.search(name): print(name[:110], row)
" "$pat"
