"""What the pair cutoff changes: NB_CUTOFF = 0.6 nm (plain truncation, the cell-list kernels) against NoCutoff (what the
reference evaluates, model.py:181-217: no setNonbondedMethod / setCutoffDistance call), same system, same start, one GPU.

  start:      dE_total, relative L2 / RMS difference of the forces
  converged:  both runs minimized to the OpenMM criterion (tolerance 10 kJ/mol/nm); the two end structures compared by
              their NoCutoff energy, radius of gyration, bond-length and nearest-non-bonded-neighbour statistics
usage: cutoff_tolerance.py [workload [n_beads]]     (run on the GPU box; writes one JSON line)"""
import json
import sys
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for


def observables(s, x):
    x = np.asarray(x, np.float64)
    rg = float(np.sqrt(((x - x.mean(0)) ** 2).sum(1).mean()))
    from multimm_amd.system import backbone_flags
    f = backbone_flags(s.n_beads, s.chr_ends)
    b = np.linalg.norm(x[1:] - x[:-1], axis=1)[(f[:-1] & 1) != 0]
    return {"rg_nm": rg, "bond_mean_nm": float(b.mean()), "bond_std_nm": float(b.std()),
            "bond_p01_nm": float(np.quantile(b, 0.01)), "bond_p99_nm": float(np.quantile(b, 0.99))}


def compare(name="chr1_50k", n_beads=None, rc=0.6, jitter=0.0):
    s_cut = synthetic_system(name, n_beads=n_beads, jitter=jitter, NB_CUTOFF=rc)
    s_all = synthetic_system(name, n_beads=n_beads, jitter=jitter, NB_CUTOFF=0.0)
    out = {"workload": s_cut.name, "n_beads": s_cut.n_beads, "cutoff_nm": rc}
    with engine_for(s_cut) as ec, engine_for(s_all) as ea:
        et_c, F_c = ec.compute()
        et_a, F_a = ea.compute()
        dF = (F_c - F_a).astype(np.float64)
        out["start"] = {
            "e_total_nocutoff": float(et_a.sum()), "dE_total": float(et_c.sum() - et_a.sum()),
            "dE_per_bead": float((et_c.sum() - et_a.sum()) / s_cut.n_beads),
            "dE_ev": float(et_c[0] - et_a[0]), "dE_gauss": float(et_c[1] - et_a[1]),
            "F_rms_nocutoff": float(np.sqrt((F_a.astype(np.float64) ** 2).sum(1).mean())),
            "dF_rms": float(np.sqrt((dF ** 2).sum(1).mean())), "dF_max": float(np.abs(dF).max()),
            "dF_rel_l2": float(np.linalg.norm(dF) / np.linalg.norm(F_a)),
        }
        st_c = ec.minimize(tolerance=10.0, max_iters=0)
        st_a = ea.minimize(tolerance=10.0, max_iters=0)
        x_c, x_a = ec.get_positions(), ea.get_positions()
        # both end structures through the reference's own energy function (NoCutoff)
        ea.set_positions(x_c)
        e_c_in_all = float(ea.compute(forces=False)[0].sum())
        ea.set_positions(x_a)
        e_a_in_all = float(ea.compute(forces=False)[0].sum())
        oc, oa = observables(s_cut, x_c), observables(s_all, x_a)
        out["converged"] = {
            "iterations_cutoff": st_c.iterations, "iterations_nocutoff": st_a.iterations,
            "status_cutoff": st_c.status, "status_nocutoff": st_a.status,
            "rms_force_cutoff": st_c.rms_force, "rms_force_nocutoff": st_a.rms_force,
            "e_nocutoff_of_cutoff_structure": e_c_in_all, "e_nocutoff_of_nocutoff_structure": e_a_in_all,
            "dE_rel": (e_c_in_all - e_a_in_all) / abs(e_a_in_all),
            "cutoff": oc, "nocutoff": oa,
            "d_rg_rel": (oc["rg_nm"] - oa["rg_nm"]) / oa["rg_nm"],
            "d_bond_mean_nm": oc["bond_mean_nm"] - oa["bond_mean_nm"],
        }
    return out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "chr1_50k"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else None
    print(json.dumps(compare(name, n)))
