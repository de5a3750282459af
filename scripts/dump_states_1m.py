"""Writes the positions of gw_1m after 150 / 1650 iterations to gpurun_out/pos1m_<n>.npy as float16-safe float32 (input of the
offline ownership comparison scripts/dd_ownership_offline.py).  usage: dump_states_1m.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
eng = engine_for(synthetic_system("gw_1m"))
done = 0
for upto in (150, 1650):
    eng.minimize(tolerance=0.0, max_iters=upto - done); done = upto
    np.save(f"gpurun_out/pos1m_{upto}.npy", eng.get_positions().astype(np.float32))
print("ok")
