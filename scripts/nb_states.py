"""Pair kernel time along a minimization: both kernels (half-shell default / full-shell deterministic) at the states
reached after 0, 10, 30, ... iterations from the Hilbert lattice.   usage: nb_states.py [workload]"""
import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED, K_CELL_BUILD
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
eng = engine_for(synthetic_system(name))
done = 0
for upto in (0, 10, 30, 60, 100, 200, 400, 1000, 2500):
    if upto > done:
        eng.set_option("deterministic", 1)
        eng.minimize(tolerance=0.0, max_iters=upto - done)
        done = upto
    t = {}
    for order in (((1, 8192), (0, 4096)), ((0, 4096), (1, 8192))) * 2:   # alternate: the first timing after a state change runs slower
        for det, variant in order:
            eng.set_option("nb_variant", variant)
            t[det] = min(t.get(det, 1e30), eng.time_kernel(K_NONBONDED, 20)[0])
    eng.set_option("nb_variant", 0)
    cb = eng.time_kernel(K_CELL_BUILD, 20)[0]
    c = eng.nb_census()
    print(f"{name} after {done:5d} iterations: full-shell {t[1]:7.1f} us, half-shell {t[0]:7.1f} us ({t[0] / t[1]:.2f}), "
          f"cell build {cb:.1f} us; pairs/bead {c['pairs_within_cutoff'] / c['n_cells'] * 0 + c['pairs_within_cutoff'] / 200000 if name == 'gw_200k' else 0:.0f}, "
          f"cells {c['n_cells']}, max/cell {c['max_per_cell']}", flush=True)
