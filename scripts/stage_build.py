"""Where the time of k_build_direct goes on one domain (-DMMX_STAGE_TIMING build, MMX_LIB=...): per role of its workgroups (work
items, bonded pass, in-cell order) the start, the end of the row prefixes and the end, from the launch's first workgroup.
usage: MMX_LIB=multimm_amd/libmmx_timing.so stage_build.py [workload=gw_200k] [iterations=25]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, load_library
lib = load_library()
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
its = int(sys.argv[2]) if len(sys.argv) > 2 else 25
s = synthetic_system(name)
NIB = 128
with engine_for(s) as e:
    e.minimize(tolerance=0.0, max_iters=its)
    t = np.zeros(8192, np.uint64)
    lib.mmx_debug_build_times(C.c_void_p(t.ctypes.data))
    cells = np.zeros(8192, np.int32)
    lib.mmx_debug_build_cells(C.c_void_p(cells.ctypes.data))
    n_own = s.n_beads
blk = t[:8190].astype(np.int64).reshape(2730, 3)
t0 = blk[:NIB, 0].min() - 100
n = int((blk[:, 0] >= t0).sum())
d = (blk - t0) / 100.0
nbr = (min((n_own + 255) // 256, 1024) + 1) // 2
print(f"{name}: {n} workgroups of the last build ({NIB} work items, {nbr} bonded, {n - NIB - nbr} order)")
for nm, a, b in (("work items", 0, NIB), ("bonded", NIB, NIB + nbr), ("order", NIB + nbr, n)):
    x = d[a:b]
    x = x[blk[a:b, 0] >= t0]
    pre = (x[:, 1] - x[:, 0]) if nm != "bonded" else np.zeros(1)
    own = x[:, 2] - x[:, 0]
    print(f"  {nm:10s} {len(x):5d} workgroups: start {x[:, 0].min():5.1f}..{x[:, 0].max():5.1f}  prefixes {pre.mean():4.1f} (max {pre.max():4.1f})"
          f"  end mean {x[:, 2].mean():5.1f} max {x[:, 2].max():5.1f}  own time mean {own.mean():5.1f} max {own.max():5.1f}")
    if nm == "order":
        print("  order workgroups, own time by index decile: " + " ".join(f"{own[i * len(own) // 10:(i + 1) * len(own) // 10].mean():.1f}" for i in range(10)))
        print("  order workgroups, END by index decile:      " + " ".join(f"{x[i * len(own) // 10:(i + 1) * len(own) // 10, 2].max():.1f}" for i in range(10)))
        cc = cells.reshape(4096, 2)[NIB + nbr:n]
        print("  population of wave 0's first cell by index decile: " + " ".join(f"{cc[i * len(own) // 10:(i + 1) * len(own) // 10, 1].mean():.0f}" for i in range(10)))
        top = np.argsort(-own)[:12]
        print("  slowest (index: us): " + ", ".join(f"{i}:{own[i]:.1f}" for i in top))
    if nm == "work items":
        print("  work-item workgroups, own time: " + " ".join(f"{v:.0f}" for v in own))
