"""Where the time of the two pair kernels goes at later states of a gw_200k minimization: diagnosis instances that stop after the
cluster cull, the j stream, the per-bead cull (nb_variant bits; results are wrong, only the timing counts).  usage: late_states.py"""
import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED
eng = engine_for(synthetic_system("gw_200k"))
eng.set_option("deterministic", 1)
done = 0
for upto in (400, 2000):
    eng.set_option("nb_variant", 0)
    eng.minimize(tolerance=0.0, max_iters=upto - done); done = upto
    for v, nm in ((8192, "full-shell"), (1024, "full-shell: cluster cull + setup + fold only"), (256 + 128, "full-shell: + stream, no bead cull, no arithmetic"), (256, "full-shell: + bead cull/ring, no arithmetic"), (4096, "half-shell"), (4096 + (16 << 16), "half-shell: cull only"), (4096 + (4 << 16), "half-shell: no arithmetic")):
        eng.set_option("nb_variant", v)
        print(done, nm, "%.1f us" % eng.time_kernel(K_NONBONDED, 20)[0], flush=True)
