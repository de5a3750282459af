import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, MMXError
for n, world in ((2050, 8), (600, 8), (300, 7)):
    s = synthetic_system("gw_200k", n_beads=n, jitter=0.02, seed=1)
    with engine_for(s) as e:
        et0, F0 = e.compute()
    try:
        engines = [engine_for(s, rank=r, world=world) for r in range(world)]
    except MMXError as ex:
        print(n, world, "create:", ex); continue
    Engine.comm_init_local(engines)
    out, err = [None] * world, []
    def work(r):
        try:
            et, f = engines[r].compute()
            st = engines[r].minimize(tolerance=0.0, max_iters=5)
            out[r] = (et, f, engines[r].owned_beads(), None, st.iterations, st.e_final)
        except Exception as ex:
            err.append((r, repr(ex)))
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in th]; [t.join(60) for t in th]
    if err or any(t.is_alive() for t in th):
        print(n, world, "errors:", err[:2], "alive:", [t.is_alive() for t in th]); continue
    F = np.zeros_like(F0)
    for et, f, lo, no, it, ef in out:
        F[lo] = f
    print(n, world, "own", [len(o[2]) for o in out], "dE", np.abs(out[0][0] - et0).max(), "dF", np.abs(F - F0).max(), "iters", out[0][4])
    for e in engines: e.close()
