"""Communication volume of a decomposed run: gw_1m on 2, 4, 8 ranks through the loopback communicator on ONE GPU -- ghosts
per rank, bytes on the wire per evaluation against the all-gather they replace, rebuilds, halts -- in the collapse phase
(first `iters` iterations from the lattice) and on the relaxed structure (`relax` iterations further).  (Timing on one GPU
says nothing about 8: only the volumes are reported.)   usage: dd_halo_stats.py [workload] [iters=150] [relax=1500] [worlds=2,4,8]"""
import os, sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_1m"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150
relax = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
worlds = [int(w) for w in (sys.argv[4] if len(sys.argv) > 4 else "2,4,8").split(",")]
KEYS = ("dd_ghosts", "dd_ghost_slots", "dd_exchanges", "dd_bytes_sent", "dd_redecompositions", "dd_sync_rebuilds", "dd_halts",
        "dd_capacity_updates", "dd_reassignments", "dd_reassign_attempts", "dd_segments_moved")
s = synthetic_system(name)
for world in worlds:
    engines = [engine_for(s, rank=r, world=world) for r in range(world)]
    for e in engines:   # DD_EVERY / DD_SKIN in the environment: lists rebuilt every K-th evaluation under a skin (default: every one, none)
        if os.environ.get("DD_EVERY"):
            e.set_option("dd_rebuild_every", float(os.environ["DD_EVERY"]))
        if os.environ.get("DD_SKIN"):
            e.set_option("dd_skin", float(os.environ["DD_SKIN"]))
        if os.environ.get("DD_SPATIAL"):   # 0: ownership stays the initial index ranges (rounds 1-3)
            e.set_option("dd_spatial", float(os.environ["DD_SPATIAL"]))
    Engine.comm_init_local(engines)
    out = [None] * world
    def work(r):
        e = engines[r]
        rows = []
        prev = {k: 0.0 for k in KEYS}
        for phase, n in (("collapse phase", iters), ("relaxed", relax), ("relaxed, next 100", 100)):
            st = e.minimize(tolerance=0.0, max_iters=n)
            g = {k: e.get_option(k) for k in KEYS}
            d = {k: g[k] - prev[k] for k in KEYS}
            d["dd_ghosts"], d["dd_ghost_slots"] = g["dd_ghosts"], g["dd_ghost_slots"]
            rows.append((phase, st.iterations, st.evaluations, st.e_final, d))
            prev = g
        out[r] = (rows, s.n_beads // world)
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    for e in engines:
        e.close()
    n_own = out[0][1]
    for p in range(3):
        rows = [o[0][p] for o in out]
        gh = [r[4]["dd_ghosts"] for r in rows]
        per = [r[4]["dd_bytes_sent"] / max(r[4]["dd_exchanges"], 1) for r in rows]
        d0 = rows[0][4]
        print(f"{name} {world} ranks, {rows[0][0]}: {rows[0][1]} iterations ({rows[0][2]} evaluations), E = {rows[0][3]:.6g}; owned/rank {n_own}; "
              f"ghosts at the end min/mean/max {min(gh):.0f}/{np.mean(gh):.0f}/{max(gh):.0f}; bytes sent per evaluation per rank "
              f"mean {np.mean(per) / 1e6:.2f} MB max {max(per) / 1e6:.2f} MB (all-gather: {16 * n_own * (world - 1) / 1e6:.1f} MB received per rank); "
              f"list rebuilds {d0['dd_redecompositions']:.0f} ({d0['dd_sync_rebuilds']:.0f} synchronous), halts {d0['dd_halts']:.0f}, "
              f"message resizes {d0['dd_capacity_updates']:.0f}; segment re-assignments {d0['dd_reassignments']:.0f} of "
              f"{d0['dd_reassign_attempts']:.0f} attempts, {d0['dd_segments_moved']:.0f} segments moved", flush=True)
