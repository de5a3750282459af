"""Randomised check of the single-domain cell build + half-shell pair kernel against the full-shell kernel: random sizes,
densities (lattice, jittered, relaxed for a random number of iterations, random gas / blob), work-item lengths, window-pass
records on/off, slot table on/off, kept cell structure on/off; every case compares forces and energies of the two kernels at
the same state, repeats the half-shell evaluation, and ends with a short minimization (which must end finite, status >= 0).
usage: n3_fuzz.py [cases=40] [seed=0]"""
import sys, dataclasses
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    n = int(rng.choice([1500, 6000, 20000, 60000, 150000, 260000]))
    kind = str(rng.choice(["lattice", "jitter", "relaxed", "gas", "blob"]))
    s = synthetic_system("gw_200k", n_beads=n, jitter=0.03 if kind == "jitter" else 0.0, seed=int(rng.integers(1 << 30)))
    if kind == "gas":
        L = (n / rng.uniform(5, 60)) ** (1 / 3) * 0.6
        s = dataclasses.replace(s, positions=rng.uniform(-L / 2, L / 2, size=(n, 3)))
    if kind == "blob":
        s = dataclasses.replace(s, positions=rng.normal(0.0, (n / 4000.0) ** (1 / 3) * 0.45, size=(n, 3)))
    opts = {"n3_long_items": int(rng.integers(0, 3)), "n3_pass_records": int(rng.integers(0, 2)), "cell_slots": int(rng.integers(0, 2)),
            "cell_reuse": int(rng.integers(0, 2)), "cell_edge_auto": int(rng.integers(0, 2))}
    relax = int(rng.integers(5, 400)) if kind == "relaxed" else 0
    with engine_for(s) as eng:
        for k, v in opts.items():
            eng.set_option(k, v)
        if relax:
            eng.minimize(tolerance=0.0, max_iters=relax)
        eng.set_option("nb_variant", 8192)
        e0, F0 = eng.compute()
        fmax = np.abs(F0).max()
        eng.set_option("nb_variant", 4096)
        worst, eworst = 0.0, 0.0
        for _ in range(3):
            e, F = eng.compute()
            worst = max(worst, float(np.abs(F - F0).max() / fmax))
            eworst = max(eworst, float(np.abs(e - e0).max() / np.abs(e0).sum()))
        items = int(eng.get_option("n3_items"))
        eng.set_option("nb_variant", 0)
        st = eng.minimize(tolerance=0.0, max_iters=int(rng.integers(20, 120)))
        fin = bool(np.isfinite(eng.get_positions()).all())
        halts = eng.get_option("cell_slot_halts") + eng.get_option("cell_stale_halts")
    ok = worst <= 3e-5 and eworst <= 3e-6 and fin and st.status >= 0 and np.isfinite(st.e_final)
    bad += 0 if ok else 1
    print(f"case {case:2d}: n={n:6d} {kind:8s} relax={relax:3d} {opts} items {items:5d}: dF/maxF {worst:.1e} dE/sumE {eworst:.1e}; "
          f"then {st.iterations} iterations, status {st.status}, halts {halts:.0f} {'ok' if ok else 'BAD'}", flush=True)
print("bad cases:", bad)
