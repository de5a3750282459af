"""Fixed cost of one mmx_minimize call: wall time of calls of K iterations (each continuing from the previous one's end),
fitted as a + b K.   usage: call_overhead.py [workload=gw_200k]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
s = synthetic_system(name)
with engine_for(s) as eng:
    eng.minimize(tolerance=0.0, max_iters=300)          # past the collapse: the per-iteration time is steady
    rows = []
    for rep in range(4):
        for K in (1, 2, 5, 10, 20, 40, 80):
            t0 = time.perf_counter()
            st = eng.minimize(tolerance=0.0, max_iters=K)
            dt = time.perf_counter() - t0
            rows.append((K, st.evaluations, dt * 1e6, st.seconds * 1e6))
    a = np.array(rows)
    for K in sorted(set(a[:, 0])):
        m = a[a[:, 0] == K]
        print(f"K={int(K):3d}: evaluations {m[:, 1].mean():5.1f}  wall {m[:, 2].mean():8.1f} us  (library's own clock {m[:, 3].mean():8.1f})  per iteration {m[:, 2].mean() / K:7.1f} us")
    A = np.stack([np.ones(len(a)), a[:, 1]], 1)
    coef = np.linalg.lstsq(A, a[:, 2], rcond=None)[0]
    print(f"fit: {coef[0]:.1f} us per call + {coef[1]:.1f} us per evaluation")
