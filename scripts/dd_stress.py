"""Randomised stress of the decomposed path through the loopback communicator (one GPU): random sizes, rank counts, rebuild
intervals, skins, message slack; every case minimizes, integrates a few MD steps, and ends with a force evaluation that is
compared with a single-domain engine at the same positions.   usage: dd_stress.py [cases=24] [seed=0]"""
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    n = int(rng.choice([1500, 4000, 9000, 20000, 60000, 130000]))
    world = int(rng.choice([2, 3, 4, 5, 8]))
    K = int(rng.choice([1, 1, 2, 5]))
    skin = float(rng.choice([0.05, 0.2, 0.4]))
    fault = int(rng.choice([0, 0, 4]))
    variant = int(rng.choice([0, 4096, 8192]))
    wl = str(rng.choice(["gw_200k", "chr1_50k"]))
    s = synthetic_system(wl, n_beads=n, jitter=0.02, seed=case)
    if -(-(-(-n // 62)) // world) * (world - 1) * 62 >= n:   # slices of whole 62-bead segments: the last rank would own nothing (the library refuses)
        print(f"case {case}: n={n} world={world}: skipped (more ranks than segment slices)")
        continue
    engines = [engine_for(s, rank=r, world=world) for r in range(world)]
    for e in engines:
        e.set_option("dd_rebuild_every", K); e.set_option("dd_skin", skin); e.set_option("inject_fault", fault); e.set_option("nb_variant", variant)
    Engine.comm_init_local(engines)
    out, err = [None] * world, []
    def work(r):
        try:
            e = engines[r]
            st = e.minimize(tolerance=0.0, max_iters=25)
            e.md_configure("langevin", dt_ps=0.002, seed=case)
            e.set_velocities_to_temperature(310.0, seed=case)
            try:
                md = e.md_step(12)
                md_ok = True
            except Exception as ex:      # a stale list during MD is reported, not repaired: allowed to fail loudly
                md_ok = "ghost list" in str(ex)
                if not md_ok: raise
            x = e.get_positions()
            e.set_positions(x)
            et, f = e.compute()
            out[r] = (st.iterations, st.status, x, et, f, e.owned_beads(), None, e.get_option("dd_halts"), md_ok)
        except Exception as ex:  # noqa: BLE001
            err.append((r, repr(ex)))
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in th]; [t.join(300) for t in th]
    tag = f"case {case}: {wl} n={n} world={world} K={K} skin={skin} fault={fault} variant={variant}"
    if err or any(t.is_alive() for t in th):
        print(tag, "FAILED", err[:2], [t.is_alive() for t in th], flush=True); bad += 1
        continue
    for e in engines:
        e.close()
    x = out[0][2]
    with engine_for(s) as ref:
        ref.set_positions(x)
        et0, F0 = ref.compute()
    F = np.zeros_like(F0)
    for o in out:
        F[o[5]] = o[4]
    ferr = np.abs(F - F0).max() / max(np.abs(F0).max(), 1e-30)
    eerr = np.abs(out[0][3] - et0).max() / max(np.abs(et0).sum(), 1e-30)
    same = all(np.array_equal(o[2], x) and o[0] == out[0][0] for o in out)
    ok = ferr <= 1e-5 and eerr <= 5e-6 and same
    bad += 0 if ok else 1
    print(tag, f"iters {out[0][0]} halts {out[0][7]:.0f} dF/maxF {ferr:.2e} dE/sumE {eerr:.2e} ranks agree {same}", "" if ok else "  <-- BAD", flush=True)
print("bad cases:", bad)
sys.exit(1 if bad else 0)
