"""Where the time of k_build_direct_dd goes (-DMMX_STAGE_TIMING build, MMX_LIB=...): per role of its workgroups (work items,
bonded pass, in-cell order) the start, the end of the row prefixes and the end, from the first workgroup's start.
usage: MMX_LIB=gpurun_out/libmmx_timing.so stage_build_dd.py [workload=gw_1m] [world=8] [relax=150]"""
import sys, threading, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, load_library, K_FORCES
lib = load_library()
name = sys.argv[1] if len(sys.argv) > 1 else "gw_1m"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
relax = int(sys.argv[3]) if len(sys.argv) > 3 else 150
s = synthetic_system(name)
engines = [engine_for(s, rank=r, world=world) for r in range(world)]
Engine.comm_init_local(engines)
def work(e):
    e.minimize(tolerance=0.0, max_iters=relax)
    e.compute()
th = [threading.Thread(target=work, args=(e,)) for e in engines]
[t.start() for t in th]; [t.join() for t in th]
NIB = 128
for r, e in enumerate(engines[:2]):
    e.set_option("dd_freeze", 1)
    e.set_option("dd_overlap", 0)
    us = e.time_kernel(K_FORCES, 5)[0]
    t = np.zeros(8192, np.uint64)
    lib.mmx_debug_build_times(C.c_void_p(t.ctypes.data))
    blk = t[:8190].astype(np.int64).reshape(2730, 3)
    live = blk[:, 0] > 0
    n = int(live.sum())
    t0 = blk[:NIB, 0].min() - 100  # (records of earlier, larger launches lie about: this launch's workgroups all start within a microsecond)
    d = (blk - t0) / 100.0
    nbr = (((e.n_own + 255) // 256) + 1) // 2
    roles = (("work items", 0, NIB), ("bonded", NIB, NIB + nbr), ("order", NIB + nbr, n))
    print(f"rank {r}: force evaluation {us:.1f} us; {n} workgroups of the build")
    for nm, a, b in roles:
        x = d[a:b]
        x = x[blk[a:b, 0] > 0]
        if len(x) == 0:
            continue
        pre = x[:, 1] - x[:, 0]
        pre = pre[x[:, 1] > 0] if nm != "bonded" else np.zeros(1)
        print(f"  {nm:10s} {len(x):5d} workgroups: start {x[:, 0].min():5.1f}..{x[:, 0].max():5.1f}  prefixes {pre.mean():4.1f} (max {pre.max():4.1f})"
              f"  end mean {x[:, 2].mean():5.1f} max {x[:, 2].max():5.1f}  own time mean {(x[:, 2] - x[:, 0]).mean():5.1f} max {(x[:, 2] - x[:, 0]).max():5.1f}")
    o = d[NIB + nbr:n]
    o = o[blk[NIB + nbr:n, 0] >= t0]
    own = o[:, 2] - o[:, 0]
    top = np.argsort(-own)[:12]
    print("  slowest order workgroups (index among them: us): " + ", ".join(f"{i}:{own[i]:.1f}" for i in top))
    print("  order workgroups, own time by index decile: " + " ".join(f"{own[i * len(own) // 10:(i + 1) * len(own) // 10].mean():.1f}" for i in range(10)))
    it = d[:NIB]
    print("  work-item workgroups, own time: " + " ".join(f"{v:.0f}" for v in (it[:, 2] - it[:, 0])))
for e in engines:
    e.close()
