"""Writes the positions of gw_200k after 60 / 400 / 2000 iterations to gpurun_out/pos_<n>.npy (input of offline analyses such as
the cluster-compactness comparison of DESIGN_HISTORY.md 5c).  usage: dump_states.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
eng = engine_for(synthetic_system("gw_200k"))
done = 0
for upto in (60, 400, 2000):
    eng.minimize(tolerance=0.0, max_iters=upto - done); done = upto
    np.save(f"gpurun_out/pos_{upto}.npy", eng.get_positions().astype(np.float32))
print("ok")
