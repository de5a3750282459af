"""Readable numpy restatement of the per-term energies (O(N^2) memory: small N only).

TEST INFRASTRUCTURE ONLY.  Used to cross-check oracle/mmx_oracle.c term by term; gradients here come
from the closed forms of SURVEY.md appendix B and are themselves checked by finite differences.
Reference call sites: see the header of mmx_oracle.c.
"""
from __future__ import annotations

import numpy as np


def energy_terms(system, positions=None, cutoff=None) -> dict:
    """Per-term energies in fp64 by direct evaluation of the reference's expressions."""
    ff = system.ff
    x = np.asarray(system.positions if positions is None else positions, dtype=np.float64)
    n = system.n_beads
    R1, R2, r_comp = system.radii
    c = system.centre
    rc = ff.NB_CUTOFF if cutoff is None else cutoff
    out = dict(ev=0.0, gauss=0.0, bond=0.0, angle=0.0, loop=0.0, container=0.0, lamina=0.0, central=0.0, chb=0.0)
    iu = np.triu_indices(n, k=1)
    if ff.EV_USE_EXCLUDED_VOLUME or ff.COB_USE_COMPARTMENT_BLOCKS or ff.SCB_USE_SUBCOMPARTMENT_BLOCKS:
        d = x[iu[0]] - x[iu[1]]
        r = np.sqrt((d * d).sum(1))
        inside = np.ones_like(r, dtype=bool) if rc <= 0 else (r < rc)
        if ff.EV_USE_EXCLUDED_VOLUME:  # "epsilon*(sigma/(r + r_small))^EV_POWER", sigma = LE_HARMONIC_BOND_R0
            e = ff.EV_EPSILON * (ff.LE_HARMONIC_BOND_R0 / (r + ff.EV_R_SMALL)) ** ff.EV_POWER
            out["ev"] = float(e[inside].sum())
        tab = system.gauss_table()
        if tab.any():  # "-E * exp(-r^2/(2*rc^2))"
            E = tab[system.labels[iu[0]].astype(int) + 2, system.labels[iu[1]].astype(int) + 2]
            e = -E * np.exp(-r * r / (2.0 * r_comp * r_comp))
            out["gauss"] = float(e[inside].sum())
    if getattr(ff, "CHB_USE_CHROMOSOMAL_BLOCKS", False):  # "E*(k_C*r^4 - r^3 + r^2); E = dE*delta(chrom1-chrom2)", all pairs
        d = x[iu[0]] - x[iu[1]]
        r = np.sqrt((d * d).sum(1))
        same = system.chrom_of[iu[0]] == system.chrom_of[iu[1]]
        out["chb"] = float((ff.CHB_DE * (ff.CHB_KC * r ** 4 - r ** 3 + r ** 2))[same].sum())
    flags = system.flags
    if ff.POL_USE_HARMONIC_BOND:
        i = np.nonzero(flags & 1)[0]
        r = np.linalg.norm(x[i] - x[i + 1], axis=1)
        out["bond"] = float((0.5 * ff.POL_HARMONIC_BOND_K * (r - ff.POL_HARMONIC_BOND_R0) ** 2).sum())
    if ff.POL_USE_HARMONIC_ANGLE:
        i = np.nonzero(flags & 2)[0]
        a, b = x[i] - x[i + 1], x[i + 2] - x[i + 1]
        cn = np.linalg.norm(np.cross(a, b), axis=1)
        th = np.arctan2(cn, (a * b).sum(1))
        out["angle"] = float((0.5 * ff.POL_HARMONIC_ANGLE_CONSTANT_K * (th - ff.POL_HARMONIC_ANGLE_R0) ** 2).sum())
    if ff.LE_USE_HARMONIC_BOND and system.n_loops:
        r = np.linalg.norm(x[system.loop_m] - x[system.loop_n], axis=1)
        out["loop"] = float((0.5 * ff.LE_HARMONIC_BOND_K * (r - system.loop_rest_lengths()) ** 2).sum())
    r = np.linalg.norm(x - c, axis=1)
    if ff.SC_USE_SPHERICAL_CONTAINER:  # "C*(max(0, r-R2)^2+max(0, R1-r)^2)"
        out["container"] = float((ff.SC_SCALE * (np.maximum(0, r - R2) ** 2 + np.maximum(0, R1 - r) ** 2)).sum())
    if ff.IBL_USE_B_LAMINA_INTERACTION:  # "B*(sin(pi*(r-R1)/(R2-R1))^8 - 1)*(delta(s+1)+delta(s+2))"
        m = system.labels < 0
        out["lamina"] = float((ff.IBL_SCALE * (np.sin(np.pi * (r[m] - R1) / (R2 - R1)) ** 8 - 1.0)).sum())
    if ff.CF_USE_CENTRAL_FORCE:  # "G*chrom_s*(r-R1)*(r-R1)"
        out["central"] = float((ff.CF_STRENGTH * system.chrom_strength * (r - R1) ** 2).sum())
    return out


def total_energy(system, positions=None, cutoff=None) -> float:
    return float(sum(energy_terms(system, positions, cutoff).values()))


def fd_forces(system, positions=None, cutoff=None, h: float = 1e-6, beads=None) -> np.ndarray:
    """Central finite-difference forces -dE/dx for the listed beads (all when None)."""
    x = np.array(system.positions if positions is None else positions, dtype=np.float64)
    beads = range(system.n_beads) if beads is None else beads
    F = np.zeros((len(list(beads)), 3))
    for bi, b in enumerate(beads):
        for k in range(3):
            xp, xm = x.copy(), x.copy()
            xp[b, k] += h
            xm[b, k] -= h
            F[bi, k] = -(total_energy(system, xp, cutoff) - total_energy(system, xm, cutoff)) / (2 * h)
    return F
