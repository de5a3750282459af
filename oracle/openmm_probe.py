"""Opportunistic OpenMM check -- TEST / BENCH INFRASTRUCTURE, like the rest of oracle/.

If (and only if) ``import openmm`` works on the machine that runs the tests or the bench, the SAME system the engine
gets is built as an ``openmm.System`` by this file -- from a ``ChromatinSystem``, with this repository's own host
code, never with files of the reference -- and evaluated / minimized on OpenMM's Reference or CPU platform:

  * ``openmm_eval``      energies per term + forces        -> compared with ``mmx_compute`` (tests/test_gpu_openmm.py)
  * ``openmm_minimize``  LocalEnergyMinimizer.minimize(...) timed -> ``cpu_baseline`` of bench.py (kind "openmm")

OpenMM is not part of this image (``import openmm`` -> ModuleNotFoundError, here and on the GPU boxes), so on this
pool both report "OpenMM unavailable" with the import error's text; the code is exercised wherever OpenMM exists.

Each Force below restates what the reference installs (energy expression, parameters, index sets); the reference
call sites are cited per term.  Default functional forms only (config.py:269-312).
"""
from __future__ import annotations

import time

import numpy as np

BEAD_MASS = 16427.889  # forcefields/ff.xml:5


def probe():
    """Returns (openmm module or None, reason)."""
    try:
        import openmm  # noqa: F401
        return openmm, "openmm %s" % getattr(openmm, "__version__", "?")
    except Exception as exc:  # ModuleNotFoundError on this image
        return None, "%s: %s" % (type(exc).__name__, exc)


_R = "r=sqrt((x-x0)^2+(y-y0)^2+(z-z0)^2)"


def build_system(s, mm):
    """``ChromatinSystem`` -> (openmm.System, [term index per Force]) in the order of add_forcefield (model.py:812-857).
    Term indices are the engine's MMX_T_*: ev, gauss, bond, angle, loop, container, lamina, central, chb."""
    from multimm_amd.system import backbone_flags
    ff = s.ff
    n = s.n_beads
    R1, R2, r_comp = s.radii
    cx, cy, cz = (float(v) for v in s.centre)
    rc = float(ff.NB_CUTOFF)
    system = mm.System()
    for _ in range(n):
        system.addParticle(BEAD_MASS)
    terms = []

    def add(force, term):
        force.setForceGroup(len(terms))
        system.addForce(force)
        terms.append(term)

    def pair_method(f):
        if rc > 0.0:  # the engine's cutoff is plain truncation = CutoffNonPeriodic without a switching function
            f.setNonbondedMethod(mm.CustomNonbondedForce.CutoffNonPeriodic)
            f.setCutoffDistance(rc)
        return f

    labels = [float(v) for v in s.labels]
    if ff.EV_USE_EXCLUDED_VOLUME:  # model.py:181-201; sigma is LE_HARMONIC_BOND_R0 (sic, model.py:175)
        if ff.EV_FORCE_TYPE != "powerlaw":
            raise NotImplementedError("openmm_probe covers the default functional forms")
        f = pair_method(mm.CustomNonbondedForce("epsilon*(sigma/(r+r_small))^power"))
        f.addGlobalParameter("epsilon", ff.EV_EPSILON)
        f.addGlobalParameter("sigma", ff.LE_HARMONIC_BOND_R0)
        f.addGlobalParameter("r_small", ff.EV_R_SMALL)
        f.addGlobalParameter("power", ff.EV_POWER)
        for _ in range(n):
            f.addParticle([])
        add(f, 0)
    if ff.COB_USE_COMPARTMENT_BLOCKS:  # model.py:231-253
        if ff.COB_FORCE_TYPE != "gaussian":
            raise NotImplementedError("openmm_probe covers the default functional forms")
        f = pair_method(mm.CustomNonbondedForce(
            "-E*exp(-r^2/(2*rcb^2));"
            "E=Ea*(delta(s1-1)+delta(s1-2))*(delta(s2-1)+delta(s2-2))+Eb*(delta(s1+1)+delta(s1+2))*(delta(s2+1)+delta(s2+2))"))
        f.addGlobalParameter("rcb", r_comp)
        f.addGlobalParameter("Ea", ff.COB_EA)
        f.addGlobalParameter("Eb", ff.COB_EB)
        f.addPerParticleParameter("s")
        for v in labels:
            f.addParticle([v])
        add(f, 1)
    if ff.SCB_USE_SUBCOMPARTMENT_BLOCKS:  # model.py:307-333
        if ff.SCB_FORCE_TYPE != "gaussian":
            raise NotImplementedError("openmm_probe covers the default functional forms")
        f = pair_method(mm.CustomNonbondedForce(
            "-E*exp(-r^2/(2*rsc^2));"
            "E=Ea1*delta(s1-2)*delta(s2-2)+Ea2*delta(s1-1)*delta(s2-1)+Eb1*delta(s1+1)*delta(s2+1)+Eb2*delta(s1+2)*delta(s2+2)"))
        f.addGlobalParameter("rsc", r_comp)
        for k, v in (("Ea1", ff.SCB_EA1), ("Ea2", ff.SCB_EA2), ("Eb1", ff.SCB_EB1), ("Eb2", ff.SCB_EB2)):
            f.addGlobalParameter(k, v)
        f.addPerParticleParameter("s")
        for v in labels:
            f.addParticle([v])
        add(f, 1)
    if ff.CHB_USE_CHROMOSOMAL_BLOCKS:  # model.py:397-419 (never truncated: the term grows with r)
        f = mm.CustomNonbondedForce("dE*(k_C*r^4-r^3+r^2)*delta(c1-c2)")
        f.addGlobalParameter("dE", ff.CHB_DE)
        f.addGlobalParameter("k_C", ff.CHB_KC)
        f.addPerParticleParameter("c")
        for v in s.chrom_of:
            f.addParticle([float(v)])
        add(f, 8)
    if ff.SC_USE_SPHERICAL_CONTAINER:  # model.py:454-466
        f = mm.CustomExternalForce("C*(max(0,r-R2)^2+max(0,R1-r)^2);" + _R)
        for k, v in (("C", ff.SC_SCALE), ("R1", R1), ("R2", R2), ("x0", cx), ("y0", cy), ("z0", cz)):
            f.addGlobalParameter(k, v)
        for i in range(n):
            f.addParticle(i, [])
        add(f, 5)
    if ff.IBL_USE_B_LAMINA_INTERACTION:  # model.py:481-507
        f = mm.CustomExternalForce("B*(sin(pi*(r-R1)/(R2-R1))^8-1)*(delta(s+1)+delta(s+2));" + _R)
        for k, v in (("B", ff.IBL_SCALE), ("R1", R1), ("R2", R2), ("pi", np.pi), ("x0", cx), ("y0", cy), ("z0", cz)):
            f.addGlobalParameter(k, v)
        f.addPerParticleParameter("s")
        for i, v in enumerate(labels):
            f.addParticle(i, [v])
        add(f, 6)
    if ff.CF_USE_CENTRAL_FORCE:  # model.py:559-586
        f = mm.CustomExternalForce("G*w*(r-R1)^2;" + _R)
        for k, v in (("G", ff.CF_STRENGTH), ("R1", R1), ("x0", cx), ("y0", cy), ("z0", cz)):
            f.addGlobalParameter(k, v)
        f.addPerParticleParameter("w")
        for i, v in enumerate(s.chrom_strength):
            f.addParticle(i, [float(v)])
        add(f, 7)
    flags = backbone_flags(n, s.chr_ends)
    if ff.POL_USE_HARMONIC_BOND:  # model.py:626-636, with its chr_ends index quirk (flags bit 0)
        f = mm.HarmonicBondForce()
        for i in np.nonzero(flags & 1)[0]:
            f.addBond(int(i), int(i) + 1, ff.POL_HARMONIC_BOND_R0, ff.POL_HARMONIC_BOND_K)
        add(f, 2)
    if ff.LE_USE_HARMONIC_BOND and s.n_loops:  # model.py:653-659
        f = mm.HarmonicBondForce()
        for m, k, r0 in zip(s.loop_m, s.loop_n, s.loop_rest_lengths()):
            f.addBond(int(m), int(k), float(r0), ff.LE_HARMONIC_BOND_K)
        add(f, 4)
    if ff.POL_USE_HARMONIC_ANGLE:  # model.py:709-720 (flags bit 1)
        f = mm.HarmonicAngleForce()
        for i in np.nonzero(flags & 2)[0]:
            f.addAngle(int(i), int(i) + 1, int(i) + 2, ff.POL_HARMONIC_ANGLE_R0, ff.POL_HARMONIC_ANGLE_CONSTANT_K)
        add(f, 3)
    return system, terms


def _context(s, mm, platform, threads=None):
    system, terms = build_system(s, mm)
    plat = mm.Platform.getPlatformByName(platform)
    props = {"Threads": str(threads)} if (threads and platform == "CPU") else {}
    ctx = mm.Context(system, mm.VerletIntegrator(0.001), plat, props)
    ctx.setPositions(np.asarray(s.positions, np.float64).tolist())
    return ctx, terms


def openmm_eval(s, platform="Reference"):
    """(energy_terms[9] kJ/mol, forces [N,3] kJ/mol/nm) of ``s`` at its positions, as getState reports them."""
    mm, why = probe()
    if mm is None:
        raise RuntimeError("OpenMM unavailable (%s)" % why)
    from openmm import unit
    ctx, terms = _context(s, mm, platform)
    et = np.zeros(9)
    for g, t in enumerate(terms):
        st = ctx.getState(getEnergy=True, groups=1 << g)
        et[t] += st.getPotentialEnergy().value_in_unit(unit.kilojoule_per_mole)
    F = ctx.getState(getForces=True).getForces(asNumpy=True).value_in_unit(unit.kilojoule_per_mole / unit.nanometer)
    return et, np.asarray(F, np.float64)


def openmm_minimize(s, max_iters, platform="CPU", threads=None):
    """Times LocalEnergyMinimizer.minimize(context, tolerance -> 0, maxIterations = max_iters) (what minimizeEnergy()
    runs, model.py:886).  Returns a dict for bench.py."""
    mm, why = probe()
    if mm is None:
        return {"available": False, "reason": why}
    from openmm import unit
    ctx, _ = _context(s, mm, platform, threads)
    e0 = ctx.getState(getEnergy=True).getPotentialEnergy().value_in_unit(unit.kilojoule_per_mole)
    t0 = time.perf_counter()
    mm.LocalEnergyMinimizer.minimize(ctx, 1e-9, int(max_iters))
    dt = time.perf_counter() - t0
    e1 = ctx.getState(getEnergy=True).getPotentialEnergy().value_in_unit(unit.kilojoule_per_mole)
    return {"available": True, "version": why, "platform": platform, "iterations": int(max_iters), "seconds": dt,
            "iters_per_s": max_iters / dt, "e_initial": e0, "e_final": e1}
