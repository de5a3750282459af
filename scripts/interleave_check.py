"""Does the pair kernel run slower when other kernels run between its launches (as in a minimization) than when it is
replayed back to back (mmx_time_kernel)?  usage: interleave_check.py [workload=gw_200k] [iterations=60]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED, K_CELL_BUILD
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
its = int(sys.argv[2]) if len(sys.argv) > 2 else 60
eng = engine_for(synthetic_system(name))
eng.minimize(tolerance=0.0, max_iters=its)
for rnd in range(3):
    back = eng.time_kernel(K_NONBONDED, 40)[0]
    inter = []
    for _ in range(40):
        eng.time_kernel(K_CELL_BUILD, 1)
        inter.append(eng.time_kernel(K_NONBONDED, 1)[0])
    single = [eng.time_kernel(K_NONBONDED, 1)[0] for _ in range(40)]
    print(f"{name} after {its}: back to back (40 per timing) {back:.1f} us; one launch per timing {np.mean(single):.1f} us; "
          f"one launch after a cell build {np.mean(inter):.1f} us (min {np.min(inter):.1f})", flush=True)
