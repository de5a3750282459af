"""Instruction budget of the half-shell pair kernel by category.  Runs k_nb_n3 with its diagnosis bits at fixed
states; under `rocprofv3 --pmc SQ_INSTS_VALU` (scripts/n3_budget.sh) the per-dispatch counter of every launch is matched to its
(state, diag) by launch order (the sequence is written to gpurun_out/n3_budget_seq.json).  Also prints the launch times.
diag: 1 no flush atomics, 2 no LDS adds and no j-side FMAs, 4 no pair arithmetic, 8 no i-side atomics, 16 cull only.
usage: n3_budget.py [workload=gw_200k] [reps=4]"""
import json, sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
DIAGS = (0, 1, 8, 2, 4, 16)
seq = []
eng = engine_for(synthetic_system(name))
done = 0
for upto in (0, 60, 400):
    if upto > done:
        eng.set_option("nb_variant", 0)
        eng.minimize(tolerance=0.0, max_iters=upto - done)
        done = upto
    for d in DIAGS:
        eng.set_option("nb_variant", 4096 + (d << 16))
        n0 = eng.get_option("n3_launches")
        t = eng.time_kernel(K_NONBONDED, reps)[0]
        seq.append({"state": done, "diag": d, "first": int(n0), "launches": int(eng.get_option("n3_launches") - n0), "us": t})
        print(f"{name} after {done:4d} iterations, diag {d:2d}: {t:7.1f} us  ({seq[-1]['launches']} launches)", flush=True)
    eng.set_option("nb_variant", 0)
json.dump({"workload": name, "seq": seq}, open("gpurun_out/n3_budget_seq.json", "w"))
eng.close()
