"""Bitwise fingerprints of `deterministic = 1` minimizations (sha256 of the positions + energies after K iterations): two
builds of the library must print the same lines.   usage: det_hash.py [iterations=120]"""
import hashlib
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for

its = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for wl, nb in (("region_5k", None), ("chr1_50k", None), ("gw_200k", 60000), ("gw_200k", None)):
    s = synthetic_system(wl, n_beads=nb)
    with engine_for(s) as eng:
        eng.set_option("deterministic", 1)
        st = eng.minimize(tolerance=0.0, max_iters=its)
        x = np.ascontiguousarray(eng.get_positions(), dtype=np.float32)
        h = hashlib.sha256(x.tobytes()).hexdigest()[:16]
        print(f"{wl} n={s.n_beads} iters={st.iterations} evals={st.evaluations} e_final={st.e_final!r} gnorm={st.gnorm_final!r} x={h}")
