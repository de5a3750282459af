"""What a Verlet skin would cost the pair kernel ("stop rebuilding the cell structure every
evaluation").  Keeping membership, clusters and work items for K evaluations needs grid cells of edge cutoff + skin; this
times the pair kernel and the cell build with cells of edge scale x cutoff (option cell_edge_scale: same results, more
candidates) along a minimization, next to the launches a kept structure would save.
usage: cell_edge_cost.py [workload=gw_200k] [scales=1.0,1.1,1.2,1.3]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED, K_CELL_BUILD
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
scales = [float(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1.0,1.1,1.2,1.3").split(",")]
eng = engine_for(synthetic_system(name))
done = 0
for upto in (0, 20, 60, 150, 400, 1000, 2500):
    if upto > done:
        eng.set_option("cell_edge_scale", 1.0)
        eng.minimize(tolerance=0.0, max_iters=upto - done)
        done = upto
    t, b, err = {}, {}, {}
    F0 = None
    for rot in range(2):
        order = scales[rot:] + scales[:rot]
        for sc in order:
            eng.set_option("cell_edge_scale", sc)
            t[sc] = min(t.get(sc, 1e30), eng.time_kernel(K_NONBONDED, 20)[0])
            b[sc] = min(b.get(sc, 1e30), eng.time_kernel(K_CELL_BUILD, 20)[0])
    for sc in scales:
        eng.set_option("cell_edge_scale", sc)
        _, F = eng.compute()
        F0 = F if F0 is None else F0
        err[sc] = np.abs(F - F0).max() / np.abs(F0).max()
    print(f"{name} after {done:5d} iterations: " + "  ".join(f"edge x{sc:.1f}: pair {t[sc]:6.1f} us, build {b[sc]:5.1f} us (dF {err[sc]:.0e})" for sc in scales), flush=True)
eng.close()
