"""Half-shell minimization from the lattice with slot rows of 64 (halts) against the clean run, by option set."""
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for
s = synthetic_system("gw_200k", n_beads=100000)
its = [int(a) for a in sys.argv[1:]] or [10, 20, 40, 80, 120]
for label, opts in (("clean", {}), ("rows64", {"inject_fault": 16}), ("rows64 legacy build", {"inject_fault": 16, "fused_build": 0}),
                    ("rows64 legacy tail", {"inject_fault": 16, "fused_tail": 0}),
                    ("rows64 legacy both", {"inject_fault": 16, "fused_tail": 0, "fused_build": 0}),
                    ("clean legacy both", {"fused_tail": 0, "fused_build": 0})):
    row = []
    for k in its:
        with engine_for(s) as eng:
            for o, v in opts.items():
                eng.set_option(o, v)
            st = eng.minimize(tolerance=0.0, max_iters=k)
            row.append((st.e_final, st.evaluations, int(eng.get_option("cell_slot_halts"))))
    print(f"{label:22s}", "  ".join(f"{e:.6e}/{ev}/{h}" for e, ev, h in row))
