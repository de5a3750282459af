"""Half-shell pair kernel with and without chunk sharing (option n3_help: a wave that finds no i-cluster left to grab joins
the one with the most 128-entry chunks of its window outstanding) along a minimization: kernel time per setting, order of
the timings rotated, minimum of three; plus the forces of every setting against the full-shell kernel at that state.
usage: n3_help_ab.py [workload=gw_200k] [settings=0,128,256,384]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
settings = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,128,256,384").split(",")]
eng = engine_for(synthetic_system(name))
done = 0
for upto in (0, 20, 60, 150, 400, 1000, 2500):
    if upto > done:
        eng.set_option("nb_variant", 0); eng.set_option("deterministic", 1)
        eng.minimize(tolerance=0.0, max_iters=upto - done)
        done = upto
    eng.set_option("deterministic", 0)
    eng.set_option("nb_variant", 8192)
    _, F0 = eng.compute()
    t_full = eng.time_kernel(K_NONBONDED, 10)[0]
    eng.set_option("nb_variant", 4096)
    t, err = {}, {}
    for rot in range(3):
        order = settings[rot % len(settings):] + settings[:rot % len(settings)]
        for hm in order:
            eng.set_option("n3_help", hm)
            t[hm] = min(t.get(hm, 1e30), eng.time_kernel(K_NONBONDED, 20)[0])
    for hm in settings:
        eng.set_option("n3_help", hm)
        _, F = eng.compute()
        err[hm] = np.abs(F - F0).max() / np.abs(F0).max()
    print(f"{name} after {done:5d} iterations: full-shell {t_full:6.1f} us | " + "  ".join(f"help>={hm}: {t[hm]:6.1f} us (dF {err[hm]:.1e})" for hm in settings), flush=True)
eng.close()
