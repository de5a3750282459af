#!/usr/bin/env python3
"""Generates tests/golden/*.npz + golden.json -- frozen input/output vectors of the hot path.

WHAT THESE ARE (read before trusting them): the reference (SFGLab/MultiMM v2.0.2) ships no golden energies,
forces or coordinates, and its arithmetic lives in OpenMM, which is not installed here (DESIGN.md section 2:
"parity unpinned").  These vectors are therefore NOT outputs of the reference.  They are the outputs of this
repository's two independent fp64 restatements of the reference's expressions -- oracle/oracle_np.py (numpy,
energies) and oracle/mmx_oracle.c (C, energies + analytic forces + L-BFGS + MD) -- frozen at the moment both
agreed with each other and the analytic forces agreed with central finite differences (asserted below before
anything is written).  They pin the oracle against silent drift and give the GPU tests a committed target.

    python tests/golden/make_golden.py        # rewrites the fixtures (review the diff before committing)
"""
import dataclasses
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from multimm_amd import synthetic_system  # noqa: E402  (input generator only)
from oracle import oracle_np  # noqa: E402
from oracle.oracle import Oracle, md_velocities, philox4x32_10  # noqa: E402

ALL_ON = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
              IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True, CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_DE=0.5)
FORMS = dict(COB_FORCE_TYPE="theta", SCB_FORCE_TYPE="yukawa", CHB_FORCE_TYPE="saturating",
             BLAMINA_FORCE_TYPE="harmonic_shell", CENTRAL_FORCE_TYPE="logistic", LE_LOOP_FORCE_TYPE="gaussian_tether")

CASES = {
    # name: (preset, kwargs of synthetic_system, minimize iterations, md steps)
    "all_terms_64_nocutoff": ("gw_200k", dict(n_beads=64, jitter=0.03, seed=3, NB_CUTOFF=0.0, **ALL_ON), 0, 0),
    "gw_512_cutoff": ("gw_200k", dict(n_beads=512, jitter=0.02, seed=1, NB_CUTOFF=0.6, **ALL_ON), 40, 10),
    "region_500_circle": ("region_500", dict(NB_CUTOFF=0.0), 30, 0),
    "forms_240_cutoff": ("gw_200k", dict(n_beads=240, jitter=0.03, seed=5, NB_CUTOFF=0.6, **ALL_ON, **FORMS), 0, 0),
}
TERMS = ("ev", "gauss", "bond", "angle", "loop", "container", "lamina", "central", "chb")


def main():
    meta = {"terms": TERMS, "cases": {}}
    for name, (preset, kw, n_min, n_md) in CASES.items():
        s = synthetic_system(preset, **kw)
        # the device path receives fp32 positions: freeze exactly those numbers
        x32 = s.positions.astype(np.float32)
        s = dataclasses.replace(s, positions=x32.astype(np.float64))
        orc = Oracle(s)  # fp32-rounded inputs, fp64 arithmetic: what the GPU tests compare with
        et, F = orc.eval()
        if not any(k.endswith("_FORCE_TYPE") and v != getattr(type(s.ff)(), k) for k, v in dataclasses.asdict(s.ff).items()):
            en = oracle_np.energy_terms(s)  # independent numpy restatement (default forms only)
            et64, _ = Oracle(s, as_float32_inputs=False).eval()
            for i, k in enumerate(TERMS):
                assert abs(et64[i] - en[k]) <= 1e-10 * max(1.0, abs(en[k])), (name, k, et64[i], en[k])
        # analytic forces against central differences of the oracle's own energy (fp64 inputs: a 1e-6 nm shift
        # does not survive the fp32 rounding `orc` applies to positions)
        orc64 = Oracle(s, as_float32_inputs=False)
        _, F64 = orc64.eval()
        assert np.abs(F64 - F).max() <= 1e-5 * np.abs(F).max()
        rng = np.random.default_rng(0)
        for b in rng.choice(s.n_beads, size=6, replace=False):
            for k in range(3):
                xp, xm = s.positions.copy(), s.positions.copy()
                xp[b, k] += 1e-6
                xm[b, k] -= 1e-6
                fd = -(orc64.energy(xp) - orc64.energy(xm)) / 2e-6
                assert abs(fd - F64[b, k]) <= 2e-5 * abs(F64[b, k]) + 2e-4 * max(1.0, np.abs(F64[b]).max()), (name, b, k)
        out = dict(positions=x32, chr_ends=s.chr_ends, labels=s.labels, loop_m=s.loop_m, loop_n=s.loop_n,
                   loop_r0=s.loop_r0, energy_terms=et, forces=F)
        if s.chrom_strength is not None:
            out["chrom_strength"] = np.asarray(s.chrom_strength, dtype=np.float64)
        case = dict(n_beads=s.n_beads, ff=dataclasses.asdict(s.ff), energy_total=float(et.sum()))
        if n_min:
            xm_, st = orc.minimize(tolerance=0.0, max_iters=n_min)
            out["minimized_positions"] = xm_
            case["minimize"] = dict(max_iters=n_min, iterations=int(st.iterations), evaluations=int(st.evaluations),
                                    status=int(st.status), e_initial=st.e_initial, e_final=st.e_final)
        if n_md:
            v0 = md_velocities(s.n_beads, 310.0, 16427.889, 7)
            xmd, vmd, mst = orc.md_step(s.positions, v0, n_md, kind="langevin", dt=0.005, temperature=310.0,
                                        friction=0.5, mass=16427.889, seed=11)
            out["md_positions"], out["md_velocities"] = xmd, vmd
            case["md"] = dict(kind="langevin", n_steps=n_md, dt=0.005, temperature=310.0, friction=0.5, mass=16427.889,
                              seed=11, velocity_seed=7, potential=mst.potential, kinetic=mst.kinetic)
            # mm.amd.AMDIntegrator (model.py:794-800) with the boost active: E above the potential energy of the start
            amd_alpha, amd_e = 300.0, float(np.floor(et.sum())) + 200.0
            xa, va, ast = orc.md_step(s.positions, v0, n_md, kind="amd", dt=0.005, mass=16427.889, amd_alpha=amd_alpha,
                                      amd_e=amd_e)
            out["amd_positions"], out["amd_velocities"] = xa, va
            case["amd"] = dict(n_steps=n_md, dt=0.005, mass=16427.889, alpha=amd_alpha, e=amd_e, velocity_seed=7,
                               potential=ast.potential, kinetic=ast.kinetic)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        meta["cases"][name] = case
        print(f"{name}: n={s.n_beads} E={et.sum():.9g}")
    # published known-answer vectors of Philox4x32-10 (Random123 kat_vectors), checked against the oracle's generator
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for c, k, w in kat:
        assert [int(v) for v in philox4x32_10(c, k)] == w
    meta["philox4x32_10"] = [dict(counter=c, key=k, out=w) for c, k, w in kat]
    # hand-derivable closed forms: independent of every line of oracle code
    meta["closed_forms"] = {
        "ev_two_beads_r0.1": {"energy": 100.0 * (0.1 / 0.15) ** 6, "force": 6 * 100.0 * (0.1 / 0.15) ** 6 / 0.15},
        "cob_AA_r0.15": {"energy": -float(np.exp(-0.5))},
        "angle_right": {"energy": 0.5 * 100.0 * (np.pi / 2) ** 2},
        "container_r_R2_plus_0.1": {"energy": 1000.0 * 0.01},
        "lamina_r_R1": {"energy": -400.0},
    }
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
