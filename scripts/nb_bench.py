"""Fixed-state timing of the pair kernel variants: lattice start and a relaxed state."""
import sys, json
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED, K_CELL_BUILD, K_BACKBONE, K_LOOPS, K_CONFINE
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
relax_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
s = synthetic_system(name)
eng = engine_for(s)
for state in ("lattice", "relaxed"):
    if state == "relaxed":
        st = eng.minimize(tolerance=0.0, max_iters=relax_iters)
    cen = eng.nb_census()
    n = s.n_beads
    print(f"[{state}] cells={cen['n_cells']} max/cell={cen['max_per_cell']} cand/bead={cen['pair_candidates']/n:.0f} "
          f"within/bead={cen['pairs_within_cutoff']/n:.0f}")
    cc = eng.cluster_census()
    print(f"   clusters={cc['n_clusters']} tiles cand={cc['tiles_candidate']:.3e} accepted={cc['tiles_accepted']:.3e} "
          f"({cc['tiles_accepted']/cc['n_clusters']:.0f}/cluster) lane-eff={cen['pairs_within_cutoff']/(64*cc['tiles_accepted']):.3f}")
    for v in variants:
        eng.set_option("nb_variant", v)
        us, by = eng.time_kernel(K_NONBONDED, 20)
        if v == 0:
            print(f"   cycles/tile/SIMD @2.1GHz = {us*1e-6*2.1e9*1024/cc['tiles_accepted']:.1f}")
        print(f"   nb_variant={v}: {us:8.1f} us  {cen['pair_candidates']/us/1e6:7.3f} Tcand/s  "
              f"{cen['pairs_within_cutoff']/us/1e6:7.3f} Tpair/s  hbm-alg {by/us/1e3:6.2f} GB/s")
    for k, nm in ((K_CELL_BUILD, "cell_build"), (K_BACKBONE, "backbone"), (K_LOOPS, "loops"), (K_CONFINE, "confine")):
        us, by = eng.time_kernel(k, 20)
        print(f"   {nm:10s}: {us:8.1f} us  alg {by/us/1e3:8.1f} GB/s")
