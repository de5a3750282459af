"""Where a decomposed rank's pair kernel spends its time: the ranks of a loopback run relax gw_1m together, then each is
frozen ("dd_freeze") and its half-shell kernel is timed ALONE from a -DMMX_N3_TIMING build (per-workgroup and per-wave
exit times), beside the number of clusters, work items and units it processed.
usage: MMX_LIB=<timing build> dd_n3_tail.py [world=8] [workload=gw_1m] [relax=150]"""
import sys, threading, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, K_NONBONDED, load_library
lib = load_library()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "gw_1m"
relax = int(sys.argv[3]) if len(sys.argv) > 3 else 150
s = synthetic_system(name)


def tail(e, label):
    diag = int(sys.argv[sys.argv.index("--diag") + 1]) if "--diag" in sys.argv else 0
    nbv = 4096 + (int(sys.argv[sys.argv.index("--nbv") + 1]) if "--nbv" in sys.argv else 0)   # e.g. 16384: the DD instance
    if "--help-min" in sys.argv:      # chunk sharing (profiles/r04_experiments/n3_chunk_sharing.patch applied: option n3_help; 0 = off)
        e.set_option("n3_help", int(sys.argv[sys.argv.index("--help-min") + 1]))
    cnt = np.zeros(16, np.uint64)
    e.set_option("nb_variant", nbv + ((diag | 128) << 16))        # a counting launch (slow: contended atomics at its end)
    lib.mmx_debug_n3_counters(C.c_void_p(cnt.ctypes.data), 1)
    e.time_kernel(K_NONBONDED, 2)
    lib.mmx_debug_n3_counters(C.c_void_p(cnt.ctypes.data), 1)
    c = cnt.astype(np.float64) / max(float(cnt[8]), 1.0)
    work = " | ".join(f"{w} i-clusters {c[o]:.0f}, candidates/cl {c[o + 1] / max(c[o], 1):.1f}, steps/cl {c[o + 2] / max(c[o], 1):.1f}, "
                      f"batches/cl {c[o + 3] / max(c[o], 1):.2f} (total {c[o + 3]:.0f})" for w, o in (("owned", 0), ("ghost", 4)))
    e.set_option("nb_variant", nbv + (diag << 16))
    t = e.time_kernel(K_NONBONDED, 5)[0]
    buf = np.zeros(512 * 20, np.uint64)
    lib.mmx_debug_n3_times(C.c_void_p(buf.ctypes.data))
    b = buf.reshape(512, 20)[:256].astype(np.int64)
    t0 = b[:, 0].min()
    wend = (b[:, 1:17] - t0) / 100.0
    bend = (b[:, 17] - t0) / 100.0
    units, later = int(b[:, 19].sum()), int(b[:, 18].sum())
    wv = np.zeros(512 * 16 * 8, np.uint32)
    lib.mmx_debug_n3_waits(C.c_void_p(wv.ctypes.data))
    wv = wv.reshape(512, 16, 8)[:256].astype(np.float64)
    ph = wv[:, :, 4:7].sum(axis=(0, 1)) / 100.0
    nst = wv[:, :, 3].sum()
    wv = wv[:, :, :3] / 100.0       # us
    nvis = max(c[0] + c[4], 1)
    life = wend - ((b[:, 0] - t0) / 100.0)[:, None]
    work += (f"\n      per wave (us): lifetime {life.mean():.1f}, in i-cluster visits {wv[:, :, 1].mean():.1f} ({100 * wv[:, :, 1].mean() / life.mean():.0f} %), "
             f"waiting for a unit / helping its flush {wv[:, :, 0].mean():.1f} ({100 * wv[:, :, 0].mean() / life.mean():.0f} %), "
             f"staging {wv[:, :, 2].mean():.1f} ({nst:.0f} units staged after the first two, {wv[:, :, 2].sum() / max(nst, 1):.1f} us each), "
             f"rest {life.mean() - wv.sum(axis=2).mean():.1f}; a visit takes {wv[:, :, 1].sum() / nvis:.1f} us: "
             f"set-up {ph[0] / nvis:.2f}, sweeps {ph[1] / nvis:.2f}, fold + i-side atomics {ph[2] / nvis:.2f}, culls and the rest {(wv[:, :, 1].sum() - ph.sum()) / nvis:.2f}")
    ncl, ncells, items = (int(e.get_option(k)) for k in ("n_clusters", "n_cells", "n3_items"))
    label += f" clusters {ncl} cells {ncells} items {items}"
    print(f"{label}: kernel {t:7.1f} us; units {units} (later passes {later}); wave exit mean {wend.mean():6.1f} max {wend.max():6.1f}; "
          f"block end p10 {np.percentile(bend, 10):6.1f} median {np.median(bend):6.1f} max {bend.max():6.1f}; "
          f"mean idle of waves before the last one ends {(wend.max() - wend).mean():5.1f} us\n      {work}", flush=True)


if world == 1:
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=relax)
        tail(eng, f"world=1 owned {s.n_beads}")
else:
    engines = [engine_for(s, rank=r, world=world) for r in range(world)]
    Engine.comm_init_local(engines)
    xs = {}
    def work(e):
        e.minimize(tolerance=0.0, max_iters=relax)
        xs[e.rank] = e.get_positions()      # a collective: every position, on every rank
        e.compute()
    th = [threading.Thread(target=work, args=(e,)) for e in engines]
    [t.start() for t in th]; [t.join() for t in th]
    x = xs[0].astype(np.float64)
    rc = 0.6    # NB_CUTOFF of the synthetic workloads
    if "--pairs" in sys.argv:
        from scipy.spatial import cKDTree
        whole = cKDTree(x)
        total = (whole.count_neighbors(whole, rc) - len(x)) // 2
        print(f"unique pairs within {rc} nm: {total / 1e6:.1f} M; an ideal rank: {total / world / 1e6:.1f} M")
        for r, e in enumerate(engines):
            own = cKDTree(x[e.owned_beads()])
            oo = (own.count_neighbors(own, rc) - e.n_own) // 2
            og = own.count_neighbors(whole, rc) - e.n_own - 2 * oo
            print(f"  rank {r}: owned-owned {oo / 1e6:6.1f} M, owned-ghost {og / 1e6:6.1f} M, evaluated {(oo + og) / 1e6:6.1f} M = "
                  f"{(oo + og) * world / total:.2f} x ideal", flush=True)
    for r, e in enumerate(engines):
        e.set_option("dd_freeze", 1)
        tail(e, f"world={world} rank={r} owned {e.n_own} ghosts {e.get_option('dd_ghosts'):.0f} slots {e.get_option('dd_ghost_slots'):.0f}")
    for e in engines:
        e.close()
