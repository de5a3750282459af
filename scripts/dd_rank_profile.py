"""Kernel profile of ONE rank of a decomposed run (rank handle without communicator: positions of the other ranks as
set by the host).  Run under rocprofv3 --kernel-trace --stats.  usage: dd_rank_profile.py [workload] [world] [rank] [reps]"""
import sys
sys.path.insert(0, ".")
import numpy as np, dataclasses
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_1m"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
s = synthetic_system(name)
with engine_for(s) as eng:
    eng.minimize(tolerance=0.0, max_iters=150)
    x = eng.get_positions()
s = dataclasses.replace(s, positions=x.astype(np.float64))
with engine_for(s, rank=rank, world=world) as eng:
    for _ in range(reps):
        eng.compute()
print("done")
