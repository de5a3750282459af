import sys, threading
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, MMXError
ALL_ON = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
              IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True)
s = synthetic_system("gw_200k", n_beads=30000, jitter=0.02, seed=3, **ALL_ON)
world = 4
engines = [engine_for(s, rank=r, world=world) for r in range(world)]
for e in engines:
    e.set_option("dd_spatial", 0)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        e.set_option(k, float(v))
Engine.comm_init_local(engines)
LOG = {r: [] for r in range(world)}
def work(e, r):
    for k in (1, 1, 1, 1, 2, 4, 10, 20, 30, 30):
        try:
            st = e.minimize(tolerance=0.0, max_iters=k)
            LOG[r].append((r, "ok", st.iterations, st.e_final, {o: e.get_option(o) for o in ("max_per_cell", "n_cells", "n_clusters", "direct_builds", "dd_halts", "dd_ghosts", "cell_slot_halts", "cell_edge")}))
        except MMXError as exc:
            LOG[r].append((r, "ERR", str(exc)[100:220], {o: e.get_option(o) for o in ("max_per_cell", "n_cells", "n_clusters", "direct_builds", "dd_halts", "kernel_error", "dd_ghosts", "cell_edge")}))
            return
th = [threading.Thread(target=work, args=(e, r)) for r, e in enumerate(engines)]
[t.start() for t in th]; [t.join() for t in th]

for r in range(world):
    for l in LOG[r]:
        print(*l)
