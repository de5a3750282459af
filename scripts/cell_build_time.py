"""Time of the whole cell-build slot (pack, scan + bonded, fill, in-cell order) at the lattice start of gw_200k; used with builds
that compile a step of k_cell_order out (DESIGN_HISTORY.md 10).  usage: [MMX_LIB=...] cell_build_time.py"""
import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_CELL_BUILD
eng = engine_for(synthetic_system("gw_200k"))
eng.set_option("profile", 0)
ts = []
for _ in range(4):
    ts.append(eng.time_kernel(K_CELL_BUILD, 30)[0])
print("cell build at the lattice: %.1f us (min of %s)" % (min(ts), ["%.1f" % t for t in ts]))
