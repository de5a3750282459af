"""Wall time of eng.minimize() against the library's own clock for the driver's window (5 warm-up iterations, then 20), as
bench.py brackets it.   usage: (cd <tree> &&) python scripts/call_wall.py [reps=6] [profile_nb]"""
import sys, time
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
s = synthetic_system("gw_200k")
for r in range(reps):
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=5)
        eng.set_option("profile", 16)
        if len(sys.argv) > 2: eng.set_option("profile_nb", int(sys.argv[2]))
        t0 = time.perf_counter()
        st = eng.minimize(tolerance=0.0, max_iters=20)
        wall = time.perf_counter() - t0
        print(f"wall {wall * 1e6:8.1f} us  library clock {st.seconds * 1e6:8.1f} us  evaluations {st.evaluations}  "
              f"-> {st.iterations / wall:7.1f} it/s by wall, {st.iterations / st.seconds:7.1f} by the library's clock", flush=True)
