"""Tail shares of the half-shell kernel (k_nb_n3, stage_unit): kernel time and agreement with the full-shell kernel for a few
tail configurations (items per workgroup in the tail x 1/2, log2(shares)) at three states of a minimization.
usage: n3_tail_ab.py [workload=gw_200k]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
eng = engine_for(synthetic_system(name))
done = 0
cfgs = [(2, 1), (4, 1), (6, 1), (2, 2), (1, 1), (3, 1), (64, 0)]
for upto in (0, 60, 400, 2000):
    if upto > done:
        eng.set_option("nb_variant", 0)
        eng.minimize(tolerance=0.0, max_iters=upto - done); done = upto
    eng.set_option("nb_variant", 8192)
    et0, F0 = eng.compute()
    best = {}
    for rnd in range(4):                       # rotate the order: the first timings after a state change run slower
        for half_items, sh in cfgs[rnd % len(cfgs):] + cfgs[:rnd % len(cfgs)]:
            eng.set_option("nb_variant", 4096 + (half_items << 24) + (sh << 28))
            et, F = eng.compute()
            err = np.abs(F - F0).max() / np.abs(F0).max()
            de = np.abs(et - et0).max() / np.abs(et0).sum()
            t = min(eng.time_kernel(K_NONBONDED, 20)[0] for _ in range(2))
            k = (half_items, sh)
            best[k] = (min(t, best[k][0]) if k in best else t, err, de)
    line = [f"({h}/2,{1 << sh}): {best[(h, sh)][0]:.1f} us [dF {best[(h, sh)][1]:.1e} dE {best[(h, sh)][2]:.1e}]" for h, sh in cfgs]
    print(f"{name} after {done}: " + "  ".join(line), flush=True)
