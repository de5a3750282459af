"""Half-shell pair kernel (default) against the full-shell kernel (option deterministic) and, for small systems,
against the fp64 oracle: energies, forces, and timing at the lattice start and after a short relaxation.
usage: n3_check.py [workload ...]      (run on the GPU box)"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED

cases = sys.argv[1:] or ["small", "chr1_50k", "gw_200k"]
for name in cases:
    if name == "small":
        from oracle.oracle import Oracle
        for nb, jit in ((512, 0.0), (4096, 0.02), (20000, 0.02)):
            s = synthetic_system("gw_200k", n_beads=nb, jitter=jit, seed=1)
            et_ref, F_ref = Oracle(s).eval()
            with engine_for(s) as eng:
                for v in (0, 4096):
                    eng.set_option("nb_variant", 4096 if v else 8192)
                    et, F = eng.compute()
                    de = np.abs(et - et_ref).max() / np.abs(et_ref).sum()
                    df = np.abs(F - F_ref).max() / np.abs(F_ref).max()
                    print(f"small n={nb} jitter={jit} variant={v}: dE/sum|E|={de:.2e} dF/max|F|={df:.2e} "
                          f"Eev={et[0]:.6f} ref {et_ref[0]:.6f} Eg={et[1]:.6f} ref {et_ref[1]:.6f}", flush=True)
        continue
    s = synthetic_system(name)
    eng = engine_for(s)
    for state in ("lattice", "relaxed"):
        if state == "relaxed":
            eng.set_option("deterministic", 1)
            eng.minimize(tolerance=0.0, max_iters=300)
        res = {}
        for v in (0, 4096):
            eng.set_option("nb_variant", 4096 if v else 8192)
            et, F = eng.compute()
            us, _ = eng.time_kernel(K_NONBONDED, 20)
            res[v] = (et, F, us)
        et0, F0, us0 = res[0]
        et1, F1, us1 = res[4096]
        print(f"{name} [{state}] full-shell {us0:.1f} us, half-shell {us1:.1f} us; "
              f"dEev={et1[0]-et0[0]:.4g} of {et0[0]:.6g}, dEg={et1[1]-et0[1]:.4g} of {et0[1]:.6g}, "
              f"max|dF|={np.abs(F1-F0).max():.3g} of max|F|={np.abs(F0).max():.4g}", flush=True)
    eng.close()
