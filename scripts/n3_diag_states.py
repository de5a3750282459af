"""Half-shell kernel with diagnosis bits along a minimization (results are WRONG with any bit but 64: timing only): what a
phase of the kernel costs at the states a run passes through.  Order of the timings rotated, minimum of three.
usage: n3_diag_states.py [workload] [diag,diag,...]   (diag: 1 no flush atomics, 2 no LDS adds, 4 no pair arithmetic, 8 no i-side atomics)"""
import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
diags = [int(d) for d in (sys.argv[2] if len(sys.argv) > 2 else "0,1,8,9").split(",")]
eng = engine_for(synthetic_system(name))
done = 0
for upto in (0, 60, 400, 2000):
    if upto > done:
        eng.set_option("nb_variant", 0)
        eng.minimize(tolerance=0.0, max_iters=upto - done)
        done = upto
    t = {}
    for rot in range(3):
        for d in diags[rot % len(diags):] + diags[:rot % len(diags)]:
            eng.set_option("nb_variant", 4096 + (d << 16))
            t[d] = min(t.get(d, 1e30), eng.time_kernel(K_NONBONDED, 20)[0])
    print(f"{name} after {done:5d} iterations: " + "  ".join(f"diag {d}: {t[d]:7.1f} us" for d in diags), flush=True)
