"""Fixed-state timing of the pair kernel variants: lattice start and (optionally) a relaxed state.
usage: nb_bench.py <workload> <relax_iters|0> <variant,variant,...> [nocensus]"""
import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED, K_CELL_BUILD, K_BACKBONE, K_LOOPS, K_CONFINE
name = sys.argv[1] if len(sys.argv) > 1 else "gw_200k"
relax_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0]
census = not (len(sys.argv) > 4 and sys.argv[4] == "nocensus")
s = synthetic_system(name)
eng = engine_for(s)
for state in ("lattice", "relaxed"):
    if state == "relaxed":
        if relax_iters <= 0:
            break
        eng.set_option("nb_variant", 0)
        eng.minimize(tolerance=0.0, max_iters=relax_iters)
    n = s.n_beads
    cen = cc = None
    if census:
        cen = eng.nb_census()
        cc = eng.cluster_census()
        sw = max(cc['beads_swept'], 1.0)
        print(f"[{state}] cells={cen['n_cells']} max/cell={cen['max_per_cell']} cand/bead={cen['pair_candidates']/n:.0f} "
              f"within/bead={cen['pairs_within_cutoff']/n:.0f} clusters={cc['n_clusters']} "
              f"tiles accepted/cluster={cc['tiles_accepted']/max(cc['n_clusters'],1):.0f} "
              f"lane-eff(tiles)={cen['pairs_within_cutoff']/(64*max(cc['tiles_accepted'],1)):.3f} "
              f"lane-eff(bead cull)={cen['pairs_within_cutoff']/(8*sw):.3f} sweep-steps/cluster={sw/64/max(cc['n_clusters'],1):.1f}")
    for v in variants:
        eng.set_option("nb_variant", v)
        us, by = eng.time_kernel(K_NONBONDED, 20)
        extra = f"  {cen['pairs_within_cutoff']/us/1e6:7.3f} Tpair/s" if cen else ""
        print(f"   nb_variant={v}: {us:8.1f} us{extra}")
    eng.set_option("nb_variant", 0)
    for k, nm in ((K_CELL_BUILD, "cell_build"), (K_BACKBONE, "backbone"), (K_LOOPS, "loops"), (K_CONFINE, "confine")):
        us, by = eng.time_kernel(k, 20)
        print(f"   {nm:10s}: {us:8.1f} us  alg {by/us/1e3:8.1f} GB/s")
