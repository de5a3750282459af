"""Per-rank kernel times of a decomposed 1M-bead system measured on ONE GPU (rank handles without communicator,
positions as set by the host): the compute side of a strong-scaling projection for BASELINE config 5.
usage: dd_projection.py [workload=gw_1m] [relax_iters=150]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED, K_CELL_BUILD, K_BACKBONE, K_LOOPS, K_CONFINE
import dataclasses

name = sys.argv[1] if len(sys.argv) > 1 else "gw_1m"
relax = int(sys.argv[2]) if len(sys.argv) > 2 else 150
s = synthetic_system(name)
with engine_for(s) as eng:
    st = eng.minimize(tolerance=0.0, max_iters=relax)
    x = eng.get_positions()
    print(f"{name}: {s.n_beads} beads, relaxed {st.iterations} iterations in {st.seconds:.2f} s "
          f"({st.iterations / st.seconds:.0f} iters/s on one GPU)")
s = dataclasses.replace(s, positions=x.astype(np.float64))
for world in (1, 2, 4, 8):
    worst = None
    for rank in sorted({0, world // 2, world - 1}):
        with engine_for(s, rank=rank, world=world) as eng:
            eng.compute()
            t = {k: eng.time_kernel(kk, 10)[0] for k, kk in (("nb", K_NONBONDED), ("build", K_CELL_BUILD),
                                                               ("backbone", K_BACKBONE), ("loops", K_LOOPS),
                                                               ("confine", K_CONFINE))}
            tot = sum(t.values())
            print(f"  world={world} rank={rank}: " + " ".join(f"{k}={v:7.1f}" for k, v in t.items()) + f"  sum={tot:7.1f} us")
            worst = max(worst or 0.0, tot)
    print(f"world={world}: slowest sampled rank {worst:.1f} us of force kernels per evaluation")
