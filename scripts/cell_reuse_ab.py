"""cell_reuse A/B ("stop rebuilding the cell structure every evaluation"): minimizations with a full
cell build per evaluation (0) and with the structure of a build kept while no bead has moved more than half the skin (1):
iterations/s over iterations 0-200 (the benchmark's window), 1000-2000 and to OpenMM's convergence criterion, how many
evaluations ran on a kept structure, how many were voided, and the exactness check: the energy the minimizer reports for its
last point against a fresh evaluation of that point.
usage: cell_reuse_ab.py [workloads=chr1_50k,gw_200k]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
factors = [float(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1.3").split(",")]
for name in (sys.argv[1] if len(sys.argv) > 1 else "chr1_50k,gw_200k").split(","):
    s = synthetic_system(name)
    for reuse, factor in [(0, 1.0)] + [(1, f) for f in factors] + [(0, 1.0)]:
        with engine_for(s) as eng:
            eng.set_option("cell_reuse", reuse)
            eng.set_option("cell_reuse_factor", factor)
            eng.minimize(tolerance=0.0, max_iters=10)
            t0 = time.perf_counter(); st = eng.minimize(tolerance=0.0, max_iters=200); early = st.iterations / (time.perf_counter() - t0)
            eng.minimize(tolerance=0.0, max_iters=790)
            t0 = time.perf_counter(); st = eng.minimize(tolerance=0.0, max_iters=1000); late = st.iterations / (time.perf_counter() - t0)
            stats = {k: eng.get_option(k) for k in ("cell_builds", "cell_reuses", "cell_stale_halts", "cell_reuse_K")}
            x = eng.get_positions()
            et, _ = eng.compute()
            exact = abs(et.sum() - st.e_final) / abs(st.e_final)
        with engine_for(s) as eng:
            eng.set_option("cell_reuse", reuse)
            eng.set_option("cell_reuse_factor", factor)
            t0 = time.perf_counter(); st = eng.minimize(tolerance=10.0, max_iters=0); dt = time.perf_counter() - t0
            st2 = {k: eng.get_option(k) for k in ("cell_builds", "cell_reuses", "cell_stale_halts")}
        print(f"{name} cell_reuse={reuse} factor={factor}: iterations 10-210 at {early:.0f} it/s, 1000-2000 at {late:.0f} it/s ({stats}); energy of the last point vs a fresh "
              f"evaluation: {exact:.1e}; to convergence: {st.iterations} iterations in {dt:.3f} s (status {st.status}, {st2})", flush=True)
