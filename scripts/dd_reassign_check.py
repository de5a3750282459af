import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 150
s = synthetic_system("gw_200k", n_beads=n, jitter=0.02, seed=3)
first = int(sys.argv[4]) if len(sys.argv) > 4 else 8
def run(spatial):
    engines = [engine_for(s, rank=r, world=world) for r in range(world)]
    for e in engines:
        e.set_option("dd_spatial", spatial); e.set_option("nb_variant", float(__import__("os").environ.get("NBV", "0"))); e.set_option("dd_reassign_first", first); e.set_option("dd_reassign_max", 4 * first)
    Engine.comm_init_local(engines)
    out = [None] * world
    def work(r):
        e = engines[r]
        st = e.minimize(tolerance=0.0, max_iters=iters)
        x = e.get_positions()
        et, f = e.compute()
        out[r] = (st, x, et, f, e.owned_beads(), {k: e.get_option(k) for k in ("dd_reassignments", "dd_segments_moved", "dd_ghosts", "max_per_cell", "n_cells")})
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    for e in engines: e.close()
    return out
for spatial in (0, 1):
    out = run(spatial)
    x = out[0][1]
    with engine_for(s) as ref:
        ref.set_positions(x); et0, F0 = ref.compute()
    F = np.zeros_like(F0)
    for o in out: F[o[4]] = o[3]
    ids = np.concatenate([o[4] for o in out])
    print("spatial", spatial, "iters", out[0][0].iterations, "E", out[0][0].e_final, "partition ok", sorted(ids.tolist()) == list(range(n)),
          "dF/max", np.abs(F - F0).max() / np.abs(F0).max(), "dE", np.abs(out[0][2] - et0).max() / np.abs(et0).sum(),
          "same x", all(np.array_equal(o[1], x) for o in out), [o[5] for o in out], "owned", [len(o[4]) for o in out])
