"""Long decomposed runs through the loopback communicator on one GPU: minimizations to OpenMM's convergence criterion on 2, 4
and 8 ranks against the single-domain run (status, iterations, final energy), with the halo statistics of the whole run.
usage: dd_soak.py [workload=gw_200k] [worlds=2,4,8] [max_iters=0]"""
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
worlds = [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8").split(",")]
max_iters = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = synthetic_system(name)
with engine_for(s) as eng:
    st0 = eng.minimize(tolerance=10.0, max_iters=max_iters)
print(f"{name} single domain: status {st0.status}, {st0.iterations} iterations, E = {st0.e_final:.6g}, rms force {st0.rms_force:.3g}", flush=True)
KEYS = ("dd_ghosts", "dd_exchanges", "dd_bytes_sent", "dd_sync_rebuilds", "dd_halts", "dd_capacity_updates", "n3_launches", "dd_reassignments",
        "dd_reassign_attempts", "dd_segments_moved")
for world in worlds:
    engines = [engine_for(s, rank=r, world=world) for r in range(world)]
    Engine.comm_init_local(engines)
    out, err = [None] * world, []
    def work(r):
        try:
            st = engines[r].minimize(tolerance=10.0, max_iters=max_iters)
            out[r] = (st.status, st.iterations, st.evaluations, st.e_final, st.rms_force, {k: engines[r].get_option(k) for k in KEYS})
        except Exception as e:  # noqa: BLE001
            err.append((r, repr(e)))
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in th]; [t.join(1000) for t in th]
    if err or any(t.is_alive() for t in th):
        print(f"{name} {world} ranks: FAILED {err[:2]} alive {[t.is_alive() for t in th]}", flush=True)
        continue
    for e in engines:
        e.close()
    assert all(o[:4] == out[0][:4] for o in out), "ranks disagree"
    st = out[0]
    g = [o[5] for o in out]
    print(f"{name} {world} ranks: status {st[0]}, {st[1]} iterations ({st[2]} evaluations), E = {st[3]:.6g} ({100 * (st[3] - st0.e_final) / abs(st0.e_final):+.2f} % "
          f"against one domain), rms force {st[4]:.3g}; halts {g[0]['dd_halts']:.0f}, synchronous rebuilds {g[0]['dd_sync_rebuilds']:.0f}, message resizes "
          f"{g[0]['dd_capacity_updates']:.0f}; mean bytes per evaluation per rank {np.mean([x['dd_bytes_sent'] / max(x['dd_exchanges'], 1) for x in g]) / 1e6:.2f} MB; "
          f"half-shell launches on rank 0: {g[0]['n3_launches']:.0f}; segment re-assignments {g[0]['dd_reassignments']:.0f} of "
          f"{g[0]['dd_reassign_attempts']:.0f} attempts ({g[0]['dd_segments_moved']:.0f} segments moved); ghosts at the end "
          f"min/mean/max {min(x['dd_ghosts'] for x in g):.0f}/{np.mean([x['dd_ghosts'] for x in g]):.0f}/{max(x['dd_ghosts'] for x in g):.0f}", flush=True)
