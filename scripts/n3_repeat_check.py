"""Repeated evaluations with the half-shell kernel against the full-shell kernel's forces (same state): catches rare protocol races
of the unit pipeline as force mismatches.   usage: n3_repeat_check.py [n_beads=30000] [repeats=60] [iterations before=0]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 60
its = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = synthetic_system("gw_200k", n_beads=nb)
with engine_for(s) as eng:
    if its:
        eng.minimize(tolerance=0.0, max_iters=its)
    eng.set_option("nb_variant", 8192)
    e0, F0 = eng.compute()
    fmax = np.abs(F0).max()
    bad = 0
    worst = 0.0
    for long_items in (0, 1):
        eng.set_option("n3_long_items", long_items)
        eng.set_option("nb_variant", 4096)
        for r in range(rep):
            e, F = eng.compute()
            err = np.abs(F - F0).max() / fmax
            worst = max(worst, err)
            if err > 2e-5:
                bad += 1
                i = np.unravel_index(np.abs(F - F0).argmax(), F.shape)
                print(f"  mismatch: items {'long' if long_items else 'short'} repeat {r}: err {err:.2e} at bead {i[0]}, {int((np.abs(F - F0).max(axis=1) > 1e-5 * fmax).sum())} beads off, dE {abs(e - e0).max():.3e}")
    print(f"{nb} beads after {its} iterations: {2 * rep} evaluations, {bad} mismatches, worst err / max|F| = {worst:.2e}")
