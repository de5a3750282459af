"""How far do trial moves go, and how long do ghost lists live under dd_adaptive?  usage: dd_adaptive_probe.py [workload] [world] [skin]"""
import sys, threading
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
skin = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
s = synthetic_system(name)
engines = [engine_for(s, rank=r, world=world) for r in range(world)]
for e in engines:
    e.set_option("dd_skin", skin)
Engine.comm_init_local(engines)
log = []
def work(e, r):
    done = 0
    for k in (100, 200, 400, 800, 1000):
        st = e.minimize(tolerance=0.0, max_iters=k)
        done += st.iterations
        if r == 0:
            log.append((done, st.evaluations, e.get_option("dd_move_seen"), e.get_option("dd_lists_serve"), e.get_option("dd_redecompositions"),
                        e.get_option("dd_halts"), e.get_option("dd_ghosts")))
th = [threading.Thread(target=work, args=(e, r)) for r, e in enumerate(engines)]
[t.start() for t in th]; [t.join() for t in th]
for l in log:
    print(f"after {l[0]:5d} iterations (+{l[1]} evaluations): largest trial move at the last poll {l[2]:.4f} nm, lists serve {l[3]:.0f} evaluations, "
          f"rebuilds so far {l[4]:.0f}, halts {l[5]:.0f}, ghosts {l[6]:.0f}")
for e in engines:
    e.close()
