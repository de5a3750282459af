"""Minimization rate with the pair kernel and its work-item length forced: engine's choice / half shell with short items / half shell
with long items / full shell.   usage: choice_items.py [workload=chr1_50k] [iterations=200,2000] [n_beads]"""
import sys, time
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "chr1_50k"
its = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "200,2000").split(",")]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else None
s = synthetic_system(name, n_beads=nb)
for n in its:
    for label, variant, long_items in (("engine's choice", 0, -1), ("half shell, 16-cluster items", 4096, 0), ("half shell, 24-cluster items", 4096, 1),
                                       ("full shell", 8192, -1)) * 2:
        with engine_for(s) as eng:
            eng.set_option("nb_variant", variant)
            eng.set_option("n3_long_items", long_items)
            eng.minimize(tolerance=0.0, max_iters=10)
            t0 = time.perf_counter()
            st = eng.minimize(tolerance=0.0, max_iters=n)
            dt = time.perf_counter() - t0
            print(f"{name}@{s.n_beads} {n:5d} iterations, {label:30s}: {st.iterations / dt:8.1f} it/s", flush=True)
