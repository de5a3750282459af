"""Where the half-shell pair kernel's wave time goes (s_memtime stamps, nb_variant diagnosis bit 128 << 16).
usage: n3_phases.py [workload] [relax_iters]"""
import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
relax = int(sys.argv[2]) if len(sys.argv) > 2 else 300
eng = engine_for(synthetic_system(name))
for state in ("lattice", "relaxed"):
    if state == "relaxed":
        eng.set_option("nb_variant", 0)
        eng.minimize(tolerance=0.0, max_iters=relax)
    eng.set_option("nb_variant", 4096)
    us0, _ = eng.time_kernel(K_NONBONDED, 20)
    eng.set_option("nb_variant", 4096 + (128 << 16))
    eng.get_option("n3_dbg7")
    reps = 10
    us, _ = eng.time_kernel(K_NONBONDED, reps)
    v = [eng.get_option("n3_dbg%d" % i) for i in range(8)]
    waves = v[4] / (reps + 1)  # time_kernel runs one warm evaluation first
    tot = sum(v[:4])
    print(f"{name} [{state}] {us0:.1f} us ({us:.1f} instrumented); waves={waves:.0f}; per wave per launch, memtime ticks (100 MHz): "
          + ", ".join(f"{n}={x / v[4]:.0f} ({100 * x / tot:.0f}%)" for n, x in zip(("compute", "stage", "barrier", "flush"), v[:4])))
