"""Robustness soak of the default path: repeated full minimizations (the half-shell kernel's float atomics make every run a
different trajectory) must all converge with status 0 to the same energy within the spread of the rugged landscape.
usage: soak.py [workload=gw_200k] [runs=6]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
s = synthetic_system(name)
es = []
for r in range(runs):
    with engine_for(s) as eng:
        t0 = time.time()
        st = eng.minimize(tolerance=10.0, max_iters=0)
        dt = time.time() - t0
        assert st.status == 0, st.status
        assert np.isfinite(st.e_final)
        es.append(st.e_final)
        print(f"{name} run {r}: status {st.status}, {st.iterations} iterations, {st.evaluations} evaluations, {dt:.2f} s, "
              f"E = {st.e_final:.6e}, rms force {st.rms_force:.2f}, half-shell launches {int(eng.get_option('n3_launches'))}", flush=True)
es = np.array(es)
print(f"{name}: {runs} runs, E spread {(es.max() - es.min()) / abs(es.mean()):.2e} relative")
