import sys, numpy as np
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
from oracle.oracle import Oracle
for n in (64, 100, 512):
    base = dict(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False)
    s = synthetic_system("chr1_50k", n_beads=n, NB_CUTOFF=0.6, **base)
    et_ref, F_ref = Oracle(s).eval()
    for v in (1, 0, 32, 64, 96):
        with engine_for(s) as eng:
            eng.set_option("nb_variant", v)
            et, F = eng.compute()
        err = np.abs(F - F_ref).max(1)
        print(n, "variant", v, "E", et[0], et_ref[0], "dE/6400=", (et[0]-et_ref[0])/6400.0*1.0, "maxFerr", err.max())
