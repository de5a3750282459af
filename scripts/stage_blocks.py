"""Per-workgroup phase times of k_tail (timing build): looks for patterns by workgroup index / XCD.  usage: MMX_LIB=... stage_blocks.py [workload] [its]"""
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
blk = t[:4096].astype(np.int64).reshape(1024, 4)
blk = blk[blk[:, 0] > 0]
t0 = blk[:, 0].min()
d = (blk - t0) / 100.0
merge = d[:, 1] - d[:, 0]; loop = d[:, 2] - d[:, 1]; epi = d[:, 3] - d[:, 2]
print("by XCD (workgroup index mod 8): merge mean/max, loop mean/max, epilogue mean")
for x in range(8):
    m = merge[x::8]; l = loop[x::8]; e = epi[x::8]
    print(f"  xcd {x}: merge {m.mean():5.2f}/{m.max():5.2f}  loop {l.mean():5.2f}/{l.max():5.2f}  epilogue {e.mean():5.2f}  start {d[x::8, 0].mean():5.2f}")
o = np.argsort(-merge)[:16]
print("slowest merges: workgroup (merge us):", ", ".join(f"{i}({merge[i]:.1f})" for i in o))
print("merge by index quartile:", [round(float(merge[i * len(merge) // 4:(i + 1) * len(merge) // 4].mean()), 2) for i in range(4)])
