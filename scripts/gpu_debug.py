import sys, numpy as np
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
from oracle.oracle import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cut = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
jit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
base = dict(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False, EV_USE_EXCLUDED_VOLUME=False)
for on in ("POL_USE_HARMONIC_BOND", "POL_USE_HARMONIC_ANGLE", "LE_USE_HARMONIC_BOND", "EV_USE_EXCLUDED_VOLUME",
           "SC_USE_SPHERICAL_CONTAINER", "COB_USE_COMPARTMENT_BLOCKS", "SCB_USE_SUBCOMPARTMENT_BLOCKS",
           "IBL_USE_B_LAMINA_INTERACTION", "CF_USE_CENTRAL_FORCE"):
    kw = dict(base); kw[on] = True
    s = synthetic_system("gw_200k", n_beads=n, jitter=jit, NB_CUTOFF=cut, **kw)
    et_ref, F_ref = Oracle(s).eval()
    with engine_for(s) as eng:
        et, F = eng.compute()
    err = np.abs(F - F_ref).max(1)
    w = np.argsort(err)[-3:][::-1]
    print(on, "E", et.sum(), et_ref.sum(), "maxFerr", err.max(), "maxF", np.abs(F_ref).max())
    for b in w:
        if err[b] > 1e-2:
            print("   bead", b, "flag", s.flags[b], "gpu", F[b], "ref", F_ref[b])
