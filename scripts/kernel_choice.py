"""End-to-end check of the per-state pair-kernel choice (use_n3, mmx_engine.hpp): wall time of a minimization from the lattice with
the choice left to the engine, with the half-shell kernel forced and with the full-shell kernel forced.
usage: kernel_choice.py [workload=gw_200k] [iterations=200,2000] [n_beads]"""
import sys, time
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
its = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "200,2000").split(",")]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else None
s = synthetic_system(name, n_beads=nb)
name = f"{name}@{s.n_beads}"
for n in its:
    for label, variant in (("engine's choice", 0), ("half shell forced", 4096), ("full shell forced", 8192)) * 2:
        with engine_for(s) as eng:
            eng.set_option("nb_variant", variant)
            eng.minimize(tolerance=0.0, max_iters=10)
            t0 = time.perf_counter()
            st = eng.minimize(tolerance=0.0, max_iters=n)
            dt = time.perf_counter() - t0
            print(f"{name} {n:5d} iterations, {label:18s}: {st.iterations / dt:8.1f} it/s  ({int(eng.get_option('n3_launches'))} half-shell launches)", flush=True)
