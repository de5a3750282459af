"""Where the time of k_tail goes (-DMMX_STAGE_TIMING build): per workgroup start / merge done / loop done / ticket, and the last
workgroup's fold + decision.   usage: MMX_LIB=<timing build> stage_times.py [workload=gw_200k] [iterations=40]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, load_library
lib = load_library()
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
its = int(sys.argv[2]) if len(sys.argv) > 2 else 40
with engine_for(synthetic_system(name)) as eng:
    eng.minimize(tolerance=0.0, max_iters=its)
    t = np.zeros(8192, np.uint64)
    lib.mmx_debug_stage_times(C.c_void_p(t.ctypes.data))
t = t.astype(np.int64)
blk = t[:4096].reshape(1024, 4)
used = blk[:, 0] > 0
blk = blk[used]
t0 = blk[:, 0].min()
us = lambda v: (v - t0) / 100.0
print(f"{name}: k_tail of the last evaluation, {used.sum()} workgroups (us from the first workgroup's start)")
for j, nm in enumerate(("start", "merge done", "loop done", "published")):
    v = us(blk[:, j])
    print(f"  {nm:11s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f}")
print(f"  folding workgroup: own partials out {us(t[4096]):6.2f}  all partials in {us(t[4098]):6.2f}  folded {us(t[4099]):6.2f}  "
      f"line search decided {us(t[4100]):6.2f}  coefficients {us(t[4101]):6.2f}  done {us(t[4097]):6.2f}")
if t[4200] > 0:
    b = t[4200:4200 + 16]
    f = lambda q: [round((int(v) - int(b[0])) / 100.0, 2) if v else None for v in b[q:q + 4]]
    print("  k_build_direct (us from workgroup 0's start; start / grid read / row prefix / done):")
    print("    workgroup 0 (totals, next grid, items)", f(0))
    print("    first order workgroup, small cells    ", f(4))
    print("    first order workgroup, large cells    ", f(8))
    print("    item workgroup 1                      ", f(12))
    if t[4216] > 0:
        g = lambda i: round((int(t[i]) - int(b[0])) / 100.0, 2)
        print(f"    first large-cell workgroup, its first cell ({int(t[4219])} beads): cell known {g(4216)}, sorted {g(4217)}, emitted {g(4218)}")
