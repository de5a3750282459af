"""Compute side of the strong-scaling projection for BASELINE config 5, measured on ONE GPU: the ranks of a decomposed
gw_1m run (loopback communicator, ghost-bead halo) relax the structure together; then every rank is frozen ("dd_freeze":
no collectives, ghosts as last received) and its kernels are timed ALONE with mmx_time_kernel -- on exactly the owned
beads and ghosts it holds in the real run.  Communication is not in these numbers.
"sum" adds the five slots as STANDALONE launches (the bonded terms are three launches of 5-6 us there); "as launched" is one
force evaluation the way the minimizer enqueues it (K_FORCES: the bonded pass rides in the cell scan's launch).
usage: dd_projection.py [workload=gw_1m] [relax_iters=150]"""
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, K_NONBONDED, K_CELL_BUILD, K_BACKBONE, K_LOOPS, K_CONFINE, K_FORCES, K_DD_LISTS

name = sys.argv[1] if len(sys.argv) > 1 else "gw_1m"
relax = int(sys.argv[2]) if len(sys.argv) > 2 else 150
s = synthetic_system(name)
SLOTS = (("nb", K_NONBONDED), ("build", K_CELL_BUILD), ("backbone", K_BACKBONE), ("loops", K_LOOPS), ("confine", K_CONFINE))
base = base_w = None
for world in (1, 2, 4, 8):
    if world == 1:
        with engine_for(s) as eng:
            st = eng.minimize(tolerance=0.0, max_iters=relax)
            print(f"{name}: {s.n_beads} beads, relaxed {st.iterations} iterations in {st.seconds:.2f} s "
                  f"({st.iterations / st.seconds:.0f} iters/s on one GPU)")
            t = {k: eng.time_kernel(kk, 10)[0] for k, kk in SLOTS}
            whole = eng.time_kernel(K_FORCES, 10)[0]
            n3 = eng.get_option("n3_launches") > 0
        rows = [(0, s.n_beads, 0, t, n3, whole)]
    else:
        engines = [engine_for(s, rank=r, world=world) for r in range(world)]
        Engine.comm_init_local(engines)
        def work(e):
            e.minimize(tolerance=0.0, max_iters=relax)
            e.compute()
        th = [threading.Thread(target=work, args=(e,)) for e in engines]
        [t.start() for t in th]; [t.join() for t in th]
        rows = []
        for r, e in enumerate(engines):     # one rank at a time, alone on the GPU
            e.set_option("dd_freeze", 1)
            t = {k: e.time_kernel(kk, 10)[0] for k, kk in SLOTS}
            rows.append((r, e.n_own, e.get_option("dd_ghosts"), t, e.get_option("n3_launches") > 0,
                         e.time_kernel(K_FORCES, 10)[0] + e.time_kernel(K_DD_LISTS, 10)[0]))
        for e in engines:
            e.close()
    worst = worst_w = 0.0
    for r, n_own, ghosts, t, n3, whole in rows:
        tot = sum(t.values())
        worst = max(worst, tot)
        worst_w = max(worst_w, whole)
        print(f"  world={world} rank={r}: owned {n_own} ghosts {ghosts:.0f} " + " ".join(f"{k}={v:7.1f}" for k, v in t.items())
              + f"  sum={tot:7.1f} us  as launched={whole:7.1f} us  pair kernel: {'half shell' if n3 else 'full shell'}")
    base = base or worst
    base_w = base_w or worst_w
    print(f"world={world}: slowest rank {worst:.1f} us of force kernels per evaluation -> {base / worst:.2f}x one rank (compute only); "
          f"as launched {worst_w:.1f} us -> {base_w / worst_w:.2f}x", flush=True)
