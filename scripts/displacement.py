"""How far beads move per iteration along a minimization (what a reused cell structure would have to tolerate).
usage: displacement.py [workload=gw_200k]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
eng = engine_for(synthetic_system(name))
done = 0
for upto in (20, 100, 200, 400, 800, 1500, 2500, 3500):
    eng.minimize(tolerance=0.0, max_iters=upto - done); done = upto
    x0 = eng.get_positions()
    cum = []
    for k in (1, 7, 8, 16):      # cumulative displacement after 1, 8, 16, 32 further iterations
        eng.minimize(tolerance=0.0, max_iters=k); done += k
        d = np.linalg.norm(eng.get_positions() - x0, axis=1)
        cum.append((d.max(), np.percentile(d, 99.9)))
    print(f"{name} at iteration {upto}: max displacement after 1 / 8 / 16 / 32 iterations: " +
          " / ".join(f"{m:.4f}" for m, _ in cum) + " nm; 99.9th percentile: " + " / ".join(f"{p:.4f}" for _, p in cum), flush=True)
