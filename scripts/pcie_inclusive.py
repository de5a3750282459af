"""PCIe-inclusive rate of the hot path: what a caller of the C ABI sees when it hands host buffers over and takes them back --
mmx_set_positions (H2D), mmx_minimize (K iterations), mmx_get_positions (D2H) -- against the minimization alone.
usage: pcie_inclusive.py [workload=gw_200k] [iterations=200]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
s = synthetic_system(name)
with engine_for(s) as eng:
    x0 = eng.get_positions().copy()
    eng.minimize(tolerance=0.0, max_iters=10)
    rows = []
    for rep in range(3):
        eng.set_positions(x0)
        eng.minimize(tolerance=0.0, max_iters=10)
        xw = eng.get_positions().copy()
        t0 = time.perf_counter(); eng.set_positions(xw); t1 = time.perf_counter()
        st = eng.minimize(tolerance=0.0, max_iters=K); t2 = time.perf_counter()
        x = eng.get_positions(); t3 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, st.iterations, st.seconds))
for up, mn, dn, it, sec in rows:
    print(f"{name}: upload {up * 1e3:.3f} ms, {it} iterations {mn * 1e3:.2f} ms wall ({sec * 1e3:.2f} ms on the library's clock), download {dn * 1e3:.3f} ms: "
          f"{it / sec:.0f} it/s resident, {it / (up + mn + dn):.0f} it/s with both transfers")
