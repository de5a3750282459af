"""Full minimization to OpenMM's default tolerance (10 kJ/mol/nm) on BASELINE configs 2 and 3."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
for name in sys.argv[1:] or ["chr1_50k", "gw_200k"]:
    s = synthetic_system(name)
    with engine_for(s) as eng:
        t0 = time.time()
        st = eng.minimize(tolerance=10.0, max_iters=0)
        dt = time.time() - t0
        d = st.as_dict()
        x = eng.get_positions()
        bl = np.linalg.norm(np.diff(x, axis=0), axis=1)
        print(f"{name}: status={st.status} iters={st.iterations} evals={st.evaluations} time={dt:.2f}s "
              f"({st.iterations/dt:.0f} it/s) E0={st.e_initial:.6g} E={st.e_final:.6g} rmsF={st.rms_force:.3g} "
              f"median bond={np.median(bl):.4f} Rg={np.sqrt(((x-x.mean(0))**2).sum(1).mean()):.3f} nm")
        print("   terms:", {k: round(v, 1) for k, v in d["energy_terms"].items()})
