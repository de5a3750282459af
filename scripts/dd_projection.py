"""Compute side of the strong-scaling projection for BASELINE config 5, measured on ONE GPU: the ranks of a decomposed
gw_1m run (loopback communicator, ghost-bead halo) relax the structure together; then every rank is frozen ("dd_freeze":
no collectives, ghosts as last received) and its kernels are timed ALONE with mmx_time_kernel -- on exactly the owned
beads and ghosts it holds in the real run.  Communication is not in these numbers.
"sum" adds the five slots as STANDALONE launches (the bonded terms are three launches of 5-6 us there); "as launched" is one
force evaluation the way the minimizer enqueues it (K_FORCES: the bonded pass rides in the cell scan's launch).
Engine options for the decomposed handles may follow as name=value (e.g. dd_rebuild_every=4 dd_skin=0.1 fused_build=0): with
lists rebuilt every K-th evaluation the list kernels are charged 1/K per evaluation ("critical path").
usage: dd_projection.py [workload=gw_1m] [relax_iters=150] [option=value ...]"""
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, K_NONBONDED, K_CELL_BUILD, K_BACKBONE, K_LOOPS, K_CONFINE, K_FORCES, K_DD_LISTS

pos = [a for a in sys.argv[1:] if "=" not in a]
opts = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
name = pos[0] if len(pos) > 0 else "gw_1m"
relax = int(pos[1]) if len(pos) > 1 else 150
worlds = tuple(int(w) for w in pos[2].split(",")) if len(pos) > 2 else (1, 2, 4, 8)
s = synthetic_system(name)
SLOTS = (("nb", K_NONBONDED), ("build", K_CELL_BUILD), ("backbone", K_BACKBONE), ("loops", K_LOOPS), ("confine", K_CONFINE))
base = base_w = None
for world in worlds:
    if world == 1:
        with engine_for(s) as eng:
            st = eng.minimize(tolerance=0.0, max_iters=relax)
            print(f"{name}: {s.n_beads} beads, relaxed {st.iterations} iterations in {st.seconds:.2f} s "
                  f"({st.iterations / st.seconds:.0f} iters/s on one GPU)")
            t = {k: eng.time_kernel(kk, 10)[0] for k, kk in SLOTS}
            whole = eng.time_kernel(K_FORCES, 10)[0]
            n3 = eng.get_option("n3_launches") > 0
        rows = [(0, s.n_beads, 0, t, n3, whole, whole, 0.0)]
    else:
        engines = [engine_for(s, rank=r, world=world) for r in range(world)]
        for e in engines:
            for k, v in opts.items():
                e.set_option(k, v)
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
            K = max(1.0, e.get_option("dd_lists_serve"))   # evaluations per set of ghost lists in force at this state
            f_us, l_us = e.time_kernel(K_FORCES, 10)[0], e.time_kernel(K_DD_LISTS, 10)[0]
            n3 = e.get_option("n3_launches") > 0
            # critical path of an evaluation on this rank, kernels only: the force evaluation as launched + the halo's own
            # kernels (list rebuild charged once per K evaluations: the message pack / unpack part of the slot is small).
            # dd_overlap (half-shell kernel): the frozen evaluation runs the list kernels, the message pack and the ghost count on
            # the second stream beside the owned beads' share of the build, exactly as a real one does -- K_FORCES is the critical
            # path then; the one-stream figure is measured next to it.
            if e.get_option("dd_overlap") > 0 and n3:
                crit = f_us
                e.set_option("dd_overlap", 0)
                serial = e.time_kernel(K_FORCES, 10)[0] + l_us / K
                e.set_option("dd_overlap", 1)
            else:
                crit = serial = f_us + l_us / K
            rows.append((r, e.n_own, e.get_option("dd_ghosts"), t, n3, crit, serial, l_us))
        for e in engines:
            e.close()
    worst = worst_w = worst_s = 0.0
    for r, n_own, ghosts, t, n3, whole, serial, l_us in rows:
        tot = sum(t.values())
        worst = max(worst, tot)
        worst_w = max(worst_w, whole)
        worst_s = max(worst_s, serial)
        print(f"  world={world} rank={r}: owned {n_own} ghosts {ghosts:.0f} " + " ".join(f"{k}={v:7.1f}" for k, v in t.items())
              + f"  sum={tot:7.1f} us  critical path={whole:7.1f} us (on one stream {serial:7.1f}, list / halo kernels {l_us:5.1f})"
              + f"  pair kernel: {'half shell' if n3 else 'full shell'}")
    base = base or worst
    base_w = base_w or worst_w
    print(f"world={world}: slowest rank {worst:.1f} us of force kernels per evaluation -> {base / worst:.2f}x one rank (compute only); "
          f"critical path (force evaluation as launched incl. list / halo kernels) {worst_w:.1f} us -> {base_w / worst_w:.2f}x"
          f" (everything on one stream: {worst_s:.1f} us -> {base_w / worst_s:.2f}x)", flush=True)
