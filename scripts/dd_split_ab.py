"""A/B on one box: the ghosts' clusters in a region of their own (option dd_split = 1: ghost clusters are never i-clusters, owned
clusters sweep the ghosts of their full stencil) against ghost clusters interleaved per cell and taking part as i-clusters
(dd_split = 0).  The ranks of a loopback run relax the workload together, then each is frozen and its cell build and pair
kernel are timed alone under both layouts, alternating.
usage: dd_split_ab.py [world=8] [workload=gw_1m] [relax=150] [--option NAME]     (NAME: another 0/1 engine option to A/B the same
way instead of dd_split, e.g. n3_pass_records)"""
import sys, threading
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, K_NONBONDED, K_CELL_BUILD

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "gw_1m"
relax = int(sys.argv[3]) if len(sys.argv) > 3 else 150
OPT = sys.argv[sys.argv.index("--option") + 1] if "--option" in sys.argv else "dd_split"
VALS = [float(v) for v in sys.argv[sys.argv.index("--values") + 1].split(",")] if "--values" in sys.argv else [1, 0]
s = synthetic_system(name)
engines = [engine_for(s, rank=r, world=world) for r in range(world)]
Engine.comm_init_local(engines)


def work(e):
    e.minimize(tolerance=0.0, max_iters=relax)
    e.compute()


th = [threading.Thread(target=work, args=(e,)) for e in engines]
[t.start() for t in th]; [t.join() for t in th]
tot = {v: [] for v in VALS}
for r, e in enumerate(engines):
    e.set_option("dd_freeze", 1)
    res = {v: [] for v in VALS}
    for split in VALS + VALS:
        e.set_option(OPT, split)
        if "--short-when-split" in sys.argv:      # short work items (16 i-clusters) under the split layout, long ones (24) without
            e.set_option("n3_long_items", 0 if split else 1)
        tb = e.time_kernel(K_CELL_BUILD, 5)[0]
        tn = e.time_kernel(K_NONBONDED, 10)[0]
        res[split].append((tb, tn, int(e.get_option("n_clusters")), int(e.get_option("n3_items"))))
    line = f"rank {r}: owned {e.n_own} ghosts {e.get_option('dd_ghosts'):.0f}"
    for split in VALS:
        tb = np.mean([x[0] for x in res[split]]); tn = np.mean([x[1] for x in res[split]])
        tot[split].append((tb, tn))
        line += f" | {OPT}={split:g}: build {tb:6.1f} pair {tn:6.1f} us (clusters {res[split][0][2]}, items {res[split][0][3]})"
    print(line, flush=True)
for split in VALS:
    a = np.array(tot[split])
    print(f"{OPT}={split:g}: pair kernel mean {a[:, 1].mean():.1f} max {a[:, 1].max():.1f} us; build mean {a[:, 0].mean():.1f} us; "
          f"slowest rank build + pair {a.sum(axis=1).max():.1f} us")
for e in engines:
    e.close()
