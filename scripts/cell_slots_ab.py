"""cell_slots A/B: trial moves write their sort keys straight into per-cell slots (no k_cell_fill launch) against the
counting sort's fill: iterations/s over iterations 10-210 (the benchmark's window) and 1000-2000, bitwise equality of the two
minimizations with `deterministic = 1` (the keys, hence the clusters, are the same).
usage: cell_slots_ab.py [workloads=chr1_50k,gw_200k]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
for name in (sys.argv[1] if len(sys.argv) > 1 else "chr1_50k,gw_200k").split(","):
    s = synthetic_system(name)
    ends = {}
    for slots in (0, 1, 0, 1):
        with engine_for(s) as eng:
            eng.set_option("cell_slots", slots)
            eng.minimize(tolerance=0.0, max_iters=10)
            t0 = time.perf_counter(); st = eng.minimize(tolerance=0.0, max_iters=200); early = st.iterations / (time.perf_counter() - t0)
            eng.minimize(tolerance=0.0, max_iters=790)
            t0 = time.perf_counter(); st = eng.minimize(tolerance=0.0, max_iters=1000); late = st.iterations / (time.perf_counter() - t0)
            halts = eng.get_option("cell_slot_halts")
        with engine_for(s) as eng:
            eng.set_option("cell_slots", slots)
            eng.set_option("deterministic", 1)
            st = eng.minimize(tolerance=0.0, max_iters=300)
            ends.setdefault(slots, (st.e_final, eng.get_positions()))
            same = st.e_final == ends[0][0] and np.array_equal(eng.get_positions(), ends[0][1])
        print(f"{name} cell_slots={slots}: iterations 10-210 at {early:.0f} it/s, 1000-2000 at {late:.0f} it/s, halts {halts:.0f}; "
              f"300 deterministic iterations end bitwise where the fill path ends: {same}", flush=True)
