"""Communication volume of a decomposed run: gw_1m on 8 (and 2, 4) ranks through the loopback communicator on ONE GPU --
the ranks' halo sizes, bytes per evaluation against the all-gather they replace, re-decompositions.  (Timing on one GPU
says nothing about 8: only the volumes are reported.)   usage: dd_halo_stats.py [workload] [iterations]"""
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_1m"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150
s = synthetic_system(name)
for world in (2, 4, 8):
    engines = [engine_for(s, rank=r, world=world) for r in range(world)]
    Engine.comm_init_local(engines)
    out = [None] * world
    def work(r):
        e = engines[r]
        st = e.minimize(tolerance=0.0, max_iters=iters)
        g = {k: e.get_option(k) for k in ("dd_ghosts", "dd_exchanges", "dd_bytes_sent", "dd_redecompositions", "dd_skin_now")}
        out[r] = (st.iterations, st.evaluations, st.e_final, g, e.n_own)
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    for e in engines:
        e.close()
    gh = [o[3]["dd_ghosts"] for o in out]
    per = [o[3]["dd_bytes_sent"] / max(o[3]["dd_exchanges"], 1) for o in out]
    n_own = out[0][4]
    print(f"{name} {world} ranks, {out[0][0]} iterations ({out[0][1]} evaluations), E = {out[0][2]:.6g}: owned/rank {n_own}; "
          f"ghosts at the end min/mean/max {min(gh):.0f}/{np.mean(gh):.0f}/{max(gh):.0f}; halo bytes sent per evaluation per rank "
          f"mean {np.mean(per) / 1e6:.2f} MB max {max(per) / 1e6:.2f} MB (all-gather: {16 * n_own * (world - 1) / 1e6:.1f} MB received per rank); "
          f"re-decompositions {out[0][3]['dd_redecompositions']:.0f}; skin now {out[0][3]['dd_skin_now']:.2f} nm", flush=True)
