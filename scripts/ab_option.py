"""Same-process A/B of one engine option on whole minimizations (kernel means from the live HIP events): the runs alternate,
each from the same start.   usage: ab_option.py <option> <valueA> <valueB> [workload=gw_200k] [iters=200] [rounds=3]"""
import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
opt, va, vb = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
name = sys.argv[4] if len(sys.argv) > 4 else "gw_200k"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 200
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 3
s = synthetic_system(name)
for r in range(rounds):
    for v in (va, vb):
        with engine_for(s) as eng:
            eng.set_option(opt, v)
            eng.minimize(tolerance=0.0, max_iters=10)
            eng.set_option("profile", 16)
            st = eng.minimize(tolerance=0.0, max_iters=iters)
            d = st.as_dict()
            print(f"{opt}={v:g}: {st.iterations / st.seconds:8.1f} it/s  e_final {st.e_final:.8g}  " +
                  " ".join(f"{k}={x:.1f}" for k, x in d["kernel_us_mean"].items() if x), flush=True)
