"""cell_edge_auto A/B: fixed-length minimizations (deterministic pair kernel excluded: default path) with grid cells of edge
cutoff throughout (0) and with cells 1.12 x wider once a poll finds < 32 beads per cutoff-sized cell (1): iterations/s over
the late part of a minimization, where the structure has thinned out, and time to OpenMM's convergence criterion.
usage: wide_cells_ab.py [workloads=chr1_50k,gw_200k]"""
import sys, time
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
for name in (sys.argv[1] if len(sys.argv) > 1 else "chr1_50k,gw_200k").split(","):
    s = synthetic_system(name)
    rows = {}
    for rep in range(2):
        for auto in (0, 1):
            with engine_for(s) as eng:
                eng.set_option("cell_edge_auto", auto)
                eng.minimize(tolerance=0.0, max_iters=1000)         # into the thinned-out phase
                t0 = time.perf_counter()
                st = eng.minimize(tolerance=0.0, max_iters=1000)
                late = st.iterations / (time.perf_counter() - t0)
                cells = eng.get_option("n_cells")
            with engine_for(s) as eng:
                eng.set_option("cell_edge_auto", auto)
                t0 = time.perf_counter()
                st = eng.minimize(tolerance=10.0, max_iters=0)
                dt = time.perf_counter() - t0
            rows.setdefault(auto, []).append((late, cells, st.iterations, dt, st.status))
    for auto in (0, 1):
        r = rows[auto]
        print(f"{name} cell_edge_auto={auto}: iterations 1000-2000 at {max(x[0] for x in r):.0f} it/s ({r[0][1]:.0f} cells); to convergence: "
              + ", ".join(f"{x[2]} iterations in {x[3]:.3f} s (status {x[4]})" for x in r), flush=True)
