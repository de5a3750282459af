"""MD throughput under engine options (same process, alternating): md_bench.py's protocol with `option=value` pairs applied.
usage: md_options_ab.py <workload> "<opt=val,opt=val;opt=val;...>" [n_steps=2000] [relax=200] [rounds=2]      (';' separates variants, '' = defaults)"""
import sys, time
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
name = sys.argv[1]
variants = sys.argv[2].split(";")
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
relax = int(sys.argv[4]) if len(sys.argv) > 4 else 200
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
s = synthetic_system(name)
for r in range(rounds):
    for var in variants:
        with engine_for(s) as eng:
            for kv in [x for x in var.split(",") if x]:
                k, v = kv.split("=")
                eng.set_option(k, float(v))
            eng.minimize(tolerance=0.0, max_iters=relax)
            eng.md_configure("langevin", dt_ps=0.001, temperature_K=310.0, friction_per_ps=0.5, seed=0)
            eng.set_velocities_to_temperature(310.0, seed=0)
            eng.md_step(20)
            t0 = time.perf_counter(); st = eng.md_step(n_steps); dt = time.perf_counter() - t0
            print(f"{name} [{var or 'defaults'}]: {n_steps / dt:.0f} steps/s ({dt / n_steps * 1e6:.1f} us/step), cells {eng.get_option('n_cells'):.0f}", flush=True)
