"""A/B of the halo beside the owned build (dd_overlap) on frozen ranks of a decomposed gw_1m run: force evaluation as launched.
usage: dd_overlap_ab.py [workload=gw_1m] [world=8] [relax=150]"""
import sys, threading
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, K_FORCES, K_DD_LISTS
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
for r, e in enumerate(engines[:3]):
    e.set_option("dd_freeze", 1)
    out = []
    for ov, go in ((0, 0), (1, 0), (1, 512), (1, 256), (1, 128), (3, 0), (3, 256), (3, 128)):
        e.set_option("dd_overlap", ov)
        e.set_option("dd_overlap_go", go)
        out.append(f"overlap={ov} go={go}: {e.time_kernel(K_FORCES, 20)[0]:6.1f}")
    print(f"rank {r}: lists {e.time_kernel(K_DD_LISTS, 20)[0]:5.1f} | " + " | ".join(out), flush=True)
for e in engines:
    e.close()
