"""One option, several values, several windows of ONE minimization each (alternating, same process): iterations/s per window and
the time to convergence.   usage: window_sweep.py <option> <v1,v2,...> [workload=gw_200k] [windows=10-210,250-500,500-750,1000-1500] [rounds=2]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
opt, vals = sys.argv[1], [float(v) for v in sys.argv[2].split(",")]
name = sys.argv[3] if len(sys.argv) > 3 else "gw_200k"
wins = [tuple(int(x) for x in w.split("-")) for w in (sys.argv[4] if len(sys.argv) > 4 else "10-210,250-500,500-750,1000-1500").split(",")]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
s = synthetic_system(name)
acc = {v: [] for v in vals}
with engine_for(s) as eng:            # a throw-away run first: the first minimization of a process is slower (code objects, allocator)
    eng.minimize(tolerance=0.0, max_iters=100)
for r in range(rounds):
    for v in vals:
        row = []
        with engine_for(s) as eng:
            eng.set_option(opt, v)
            done = 0
            for a, b in wins:
                if a > done:
                    eng.minimize(tolerance=0.0, max_iters=a - done)
                t0 = time.perf_counter(); st = eng.minimize(tolerance=0.0, max_iters=b - a); row.append(st.iterations / (time.perf_counter() - t0))
                done = b
        with engine_for(s) as eng:
            eng.set_option(opt, v)
            t0 = time.perf_counter(); st = eng.minimize(tolerance=10.0, max_iters=0); row.append(time.perf_counter() - t0); row.append(st.iterations)
        acc[v].append(row)
for v in vals:
    a = np.array(acc[v]).mean(axis=0)
    print(f"{name} {opt}={v:g}: " + "  ".join(f"{w[0]}-{w[1]}: {x:7.1f} it/s" for w, x in zip(wins, a)) + f"  | to convergence {a[-2]:.3f} s ({a[-1]:.0f} iterations)", flush=True)
