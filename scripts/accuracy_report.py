"""Per-term accuracy of the device path against the fp64 oracle on a relaxed state (test infrastructure: imports
oracle/).  usage: accuracy_report.py [workload=gw_200k] [n_beads=0] [relax_iters=300]"""
import sys, dataclasses
sys.path.insert(0, ".")
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, TERM_NAMES
from oracle.oracle import Oracle

name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
relax = int(sys.argv[3]) if len(sys.argv) > 3 else 300
extra = dict(SCB_USE_SUBCOMPARTMENT_BLOCKS=True, CF_USE_CENTRAL_FORCE=True) if "all" in sys.argv else {}
s = synthetic_system(name, n_beads=nb or None, **extra)
with engine_for(s) as eng:
    eng.minimize(tolerance=0.0, max_iters=relax)
    x = eng.get_positions().astype(np.float64)
    s2 = dataclasses.replace(s, positions=x)
with engine_for(s2) as eng:
    et, F = eng.compute()
et_ref, F_ref = Oracle(s2).eval()
print(f"{name}: {s.n_beads} beads after {relax} iterations; sum|E_t| = {np.abs(et_ref).sum():.6g} kJ/mol")
for t, nm in enumerate(TERM_NAMES):
    if et_ref[t] != 0.0 or et[t] != 0.0:
        print(f"  {nm:10s} ref {et_ref[t]:18.4f}  gpu-ref {et[t] - et_ref[t]:12.5f}  rel {(et[t] - et_ref[t]) / abs(et_ref[t]):9.2e}")
err = np.abs(F.astype(np.float64) - F_ref)
fn = np.linalg.norm(F_ref, axis=1)
print(f"  forces: max |F| {np.abs(F_ref).max():.2f}, rms |F_i| {np.sqrt((fn ** 2).mean()):.2f}; max abs err {err.max():.3e} "
      f"({err.max() / np.abs(F_ref).max():.2e} of max), rms err {np.sqrt((err ** 2).mean()):.3e} "
      f"({np.sqrt((err ** 2).sum() / (F_ref ** 2).sum()):.2e} relative L2)")
