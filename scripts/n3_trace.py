"""Event trace of the half-shell kernel's unit pipeline (first 32 workgroups of one launch, -DMMX_N3_TIMING build): every i-cluster
visit (unit, wave, start, end, batches), every wait at the entry of a unit, staging, flush-job open and window-ready times.
Prints a per-unit table and a per-wave timeline of workgroup 0, saves the raw events (gpurun_out/n3_trace_<state>.npz).
usage: MMX_LIB=<timing build> n3_trace.py [workload=gw_200k] [relax iterations=150]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED, load_library
lib = load_library()
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
relax = int(sys.argv[2]) if len(sys.argv) > 2 else 150
NB, NE = 32, 1024
with engine_for(synthetic_system(name)) as eng:
    if relax:
        eng.minimize(tolerance=0.0, max_iters=relax)
    eng.set_option("nb_variant", 4096)
    import os
    if os.environ.get("N3_SPLIT"):
        eng.set_option("n3_split", int(os.environ["N3_SPLIT"]))
    t = eng.time_kernel(K_NONBONDED, 3)[0]
    ev = np.zeros(NB * NE * 8, np.uint32)
    cnt = np.zeros(NB, np.uint32)
    lib.mmx_debug_n3_trace(C.c_void_p(ev.ctypes.data), C.c_void_p(cnt.ctypes.data))
ev = ev.reshape(NB, NE, 8)
np.savez_compressed(f"gpurun_out/n3_trace_{name}_{relax}.npz", ev=ev, cnt=cnt, kernel_us=t)
print(f"{name} after {relax} iterations: kernel {t:.1f} us; events per workgroup {cnt.min()}..{cnt.max()}")
tot = {"visit": 0.0, "wait": 0.0}
slack = []
for b in range(NB):
    e = ev[b, :min(cnt[b], NE)].astype(np.int64)
    kind, wave = e[:, 0] & 255, e[:, 0] >> 8
    t0 = e[:, 3].min()
    vis = e[kind == 0]
    tot["visit"] += (vis[:, 4] - vis[:, 3]).sum() / 100.0
    w = e[kind == 1]
    tot["wait"] += (w[:, 4] - w[:, 3]).sum() / 100.0
    if b == 0:
        print("workgroup 0, per unit: visits, duration min/mean/max us, batches min/mean/max, first start, last end, job open, next-but-one ready")
        units = np.unique(vis[:, 1])
        for u in units:
            v = vis[vis[:, 1] == u]
            d = (v[:, 4] - v[:, 3]) / 100.0
            jo = e[(kind == 3) & (e[:, 1] == u)]
            rd = e[(kind == 4) & (e[:, 1] == u + 2)]
            print(f"  unit {u:3d}: {len(v):2d} visits, {d.min():5.1f}/{d.mean():5.1f}/{d.max():5.1f} us, batches {v[:, 5].min():3d}/{v[:, 5].mean():5.1f}/{v[:, 5].max():3d}, "
                  f"start {(v[:, 3].min() - t0) / 100.0:6.1f}..{(v[:, 3].max() - t0) / 100.0:6.1f}, end {(v[:, 4].min() - t0) / 100.0:6.1f}..{(v[:, 4].max() - t0) / 100.0:6.1f}, "
                  f"job open {(jo[0, 3] - t0) / 100.0 if len(jo) else -1:6.1f}, unit {u + 2} ready {(rd[0, 3] - t0) / 100.0 if len(rd) else -1:6.1f}")
        print("workgroup 0, per wave: unit:i-cluster[start-end]  (w = wait)")
        for wv in range(16):
            row = []
            sel = e[(wave == wv) & (kind <= 1)]
            for x in sel[np.argsort(sel[:, 3])]:
                if (x[0] & 255) == 0:
                    row.append(f"{x[1]}:{x[2]}[{(x[3] - t0) / 100.0:.0f}-{(x[4] - t0) / 100.0:.0f}]")
                elif x[4] - x[3] > 100:
                    row.append(f"w{x[1]}({(x[4] - x[3]) / 100.0:.0f})")
            print(f"  wave {wv:2d}: " + " ".join(row))
print(f"first {NB} workgroups: visits {tot['visit'] / NB / 16:.1f} us per wave, waits {tot['wait'] / NB / 16:.1f} us per wave")
