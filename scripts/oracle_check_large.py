"""Both pair kernels against the fp64 oracle at sizes where the half-shell kernel's work items, tail shares and multi-pass
windows are all in play (test infrastructure: imports oracle/).  usage: oracle_check_large.py [n_beads ...]"""
import sys, dataclasses
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
from oracle.oracle import Oracle
ALL_ON = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
              IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True)
for n in [int(v) for v in sys.argv[1:]] or [50000, 200000]:
    s = synthetic_system("gw_200k", n_beads=n, **ALL_ON)
    x = s.positions
    done = 0
    for relax in (0, 60, 400):
        if relax:
            with engine_for(dataclasses.replace(s, positions=x)) as eng:   # relax further on the default path
                eng.minimize(tolerance=0.0, max_iters=relax - done)
                x = eng.get_positions().astype(np.float64)
            done = relax
        # a fresh system at these positions: the mass centre of the confinement terms belongs to the system description
        s2 = dataclasses.replace(s, positions=x)
        et_ref, F_ref = Oracle(s2).eval()
        with engine_for(s2) as eng:
            for nm, v in (("half shell", 4096), ("full shell", 8192)):
                eng.set_option("nb_variant", v)
                et, F = eng.compute()
                de = np.abs(et - et_ref).max() / np.abs(et_ref).sum()
                df = np.abs(F - F_ref).max() / np.abs(F_ref).max()
                l2 = np.sqrt(((F - F_ref) ** 2).sum() / (F_ref ** 2).sum())
                print(f"n={n} after {relax:3d} iterations, {nm}: max |dE_term| / sum|E| = {de:.1e}, max |dF| / max|F| = {df:.1e}, rel L2 = {l2:.1e}", flush=True)
                # the lattice start has thousands of pairs exactly at the cutoff (tests/test_gpu_parity.py, *_AT_CUTOFF bands)
                assert (de < 3e-5 and df < 1e-5) if relax == 0 else (de < 2e-6 and df < 4e-6), (de, df)
