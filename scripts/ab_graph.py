"""hipGraph replay of the minimizer's evaluations against launch-by-launch submission, same process, same box.
usage: ab_graph.py [workload ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
GE = int(__import__('os').environ.get('GRAPH_EVALS', '2'))
for name in (sys.argv[1:] or ["region_5k", "chr1_50k", "gw_200k"]):
    s = synthetic_system(name)
    out = {}
    for g in (0, 1, 0, 1):
        with engine_for(s) as eng:
            eng.set_option("use_graph", g)
            eng.set_option("graph_evals", GE)
            eng.set_option("deterministic", 1)
            eng.minimize(tolerance=0.0, max_iters=10)
            t0 = time.perf_counter()
            st = eng.minimize(tolerance=0.0, max_iters=400)
            dt = time.perf_counter() - t0
            out.setdefault(g, []).append((st.iterations / dt, st.e_final, st.evaluations))
    same = out[0][0][1:] == out[1][0][1:]
    print(f"{name}: direct {out[0][0][0]:.0f} / {out[0][1][0]:.0f} it/s, graph {out[1][0][0]:.0f} / {out[1][1][0]:.0f} it/s; "
          f"bitwise-equal results: {same}", flush=True)
