import sys, numpy as np
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
from oracle.oracle import Oracle
for n in (8, 16, 64, 512):
    base = dict(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False)
    s = synthetic_system("chr1_50k", n_beads=n, NB_CUTOFF=0.6, **base)
    et_ref, F_ref = Oracle(s).eval()
    for v in (2, 0):
        with engine_for(s) as eng:
            eng.set_option("nb_variant", v)
            et, F = eng.compute()
        err = np.abs(F - F_ref).max(1)
        print(n, "variant", v, "E", et[0], et_ref[0], "maxFerr", err.max(), "bad beads", np.nonzero(err > 1e-1)[0][:20])
