"""MD throughput: steps/s of the Langevin integrator on the same force kernels.
usage: md_bench.py [workload=gw_200k] [n_steps=1000] [relax_iters=200]"""
import sys
import time
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for

name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
relax = int(sys.argv[3]) if len(sys.argv) > 3 else 200
s = synthetic_system(name)
with engine_for(s) as eng:
    eng.minimize(tolerance=0.0, max_iters=relax)
    eng.md_configure("langevin", dt_ps=0.001, temperature_K=310.0, friction_per_ps=0.5, seed=0)
    eng.set_velocities_to_temperature(310.0, seed=0)
    eng.md_step(20)  # warm
    t0 = time.perf_counter()
    st = eng.md_step(n_steps)
    dt = time.perf_counter() - t0
    print(f"{name}: {s.n_beads} beads, {n_steps} Langevin steps in {dt:.3f} s = {n_steps / dt:.0f} steps/s "
          f"({dt / n_steps * 1e3:.3f} ms/step); T = {st.temperature:.1f} K, E_pot = {st.potential:.6g} kJ/mol")
