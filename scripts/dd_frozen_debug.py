"""Why are some frozen ranks slow?  usage: dd_frozen_debug.py [world=8] [relax=150] [spatial=1]"""
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, K_NONBONDED, K_CELL_BUILD
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
relax = int(sys.argv[2]) if len(sys.argv) > 2 else 150
spatial = float(sys.argv[3]) if len(sys.argv) > 3 else 1
s = synthetic_system("gw_1m")
engines = [engine_for(s, rank=r, world=world) for r in range(world)]
for e in engines: e.set_option("dd_spatial", spatial)
Engine.comm_init_local(engines)
def work(e):
    e.minimize(tolerance=0.0, max_iters=relax)
    e.compute()
th = [threading.Thread(target=work, args=(e,)) for e in engines]
[t.start() for t in th]; [t.join() for t in th]
for r, e in enumerate(engines):
    e.set_option("dd_freeze", 1)
    info0 = {k: e.get_option(k) for k in ("n_clusters", "n_cells", "n3_items", "max_per_cell", "order_fallbacks", "kernel_error", "dd_ghosts", "dd_ghost_slots")}
    t_nb = e.time_kernel(K_NONBONDED, 5)[0]
    info = {k: e.get_option(k) for k in ("n_clusters", "n_cells", "n3_items", "max_per_cell", "order_fallbacks", "kernel_error")}
    e.set_option("nb_variant", 8192)
    t_full = e.time_kernel(K_NONBONDED, 5)[0]
    e.set_option("nb_variant", 0)
    t_b = e.time_kernel(K_CELL_BUILD, 5)[0]
    x = None
    print(f"rank {r}: owned {e.n_own} nb {t_nb:.1f} full-shell {t_full:.1f} build {t_b:.1f} before {info0} after {info}", flush=True)
for e in engines: e.close()
