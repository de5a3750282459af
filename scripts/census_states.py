import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
for name in ("gw_200k", "gw_1m"):
    s = synthetic_system(name)
    eng = engine_for(s)
    done = 0
    for upto in (0, 60, 150, 400, 2000):
        if upto > done:
            eng.minimize(tolerance=0.0, max_iters=upto - done); done = upto
        c = eng.nb_census(); cc = eng.cluster_census()
        print(f"{name} after {done}: directed pairs within r_c per bead {c['pairs_within_cutoff']/s.n_beads:.0f}, cells {c['n_cells']}, max/cell {c['max_per_cell']}, "
              f"beads/cell {s.n_beads/c['n_cells']:.0f}, swept lane fraction in cutoff {c['pairs_within_cutoff']/(8*max(cc['beads_swept'],1)):.3f}", flush=True)
    eng.close()
