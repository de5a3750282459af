"""Same-process A/B of SEVERAL values of one engine option on whole minimizations (kernel means from the live HIP events): the
runs alternate, each from the same start.   usage: ab_values.py <option> <v1,v2,...> [workload=gw_200k] [iters=200] [rounds=3] [warm=10]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
opt, vals = sys.argv[1], [float(v) for v in sys.argv[2].split(",")]
name = sys.argv[3] if len(sys.argv) > 3 else "gw_200k"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
warm = int(sys.argv[6]) if len(sys.argv) > 6 else 10
s = synthetic_system(name)
acc = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        with engine_for(s) as eng:
            eng.set_option(opt, v)
            eng.minimize(tolerance=0.0, max_iters=warm)
            eng.set_option("profile", 16)
            st = eng.minimize(tolerance=0.0, max_iters=iters)
            d = st.as_dict()
            acc[v].append((st.iterations / st.seconds, d["kernel_us_mean"].get("nonbonded", 0.0)))
for v in vals:
    a = np.array(acc[v])
    print(f"{name} iterations {warm}-{warm + iters}  {opt}={v:g}: {a[:, 0].mean():8.1f} it/s (min {a[:, 0].min():.1f} max {a[:, 0].max():.1f})  "
          f"pair kernel {a[:, 1].mean():.1f} us", flush=True)
