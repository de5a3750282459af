"""Robustness at sizes beyond BASELINE's largest config: a 4 M-bead genome-wide system on one GPU.
usage: big_system.py [n_beads=4000000] [iters=30]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t0 = time.time()
s = synthetic_system("gw_1m", n_beads=n)
print(f"system: {n} beads, {s.n_loops} loops, built in {time.time() - t0:.1f} s", flush=True)
with engine_for(s) as eng:
    et, F = eng.compute()
    print("energies:", {k: float(f"{v:.6g}") for k, v in zip(("ev", "gauss", "bond", "angle", "loop", "cont", "lam"), et)},
          "max|F| =", float(np.abs(F).max()), flush=True)
    st = eng.minimize(tolerance=0.0, max_iters=iters)
    print(f"minimize: {st.iterations} iterations, {st.evaluations} evaluations in {st.seconds:.2f} s "
          f"({st.iterations / st.seconds:.1f} it/s), E {st.e_initial:.6g} -> {st.e_final:.6g}, status {st.status}, "
          f"order_fallbacks {eng.get_option('order_fallbacks'):.0f}")
    assert np.isfinite(st.e_final) and st.e_final < st.e_initial
