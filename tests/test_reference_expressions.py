"""The fp64 oracle's energy expressions against the REFERENCE'S OWN expression strings.

The arithmetic of the reference's force field lives in strings it hands to OpenMM (`model.py:164-720`: the argument of
`mm.Custom*Force(...)` / `setEnergyFunction(...)`, per `*_FORCE_TYPE` branch) plus the radii of `set_radiuses`
(`model.py:1016-1067`).  scripts/make_reference_energy_fixture.py reads those strings, the parameter names and the source text
of the parameter values from the reference's source with `ast` (nothing is imported: the module needs OpenMM) into
tests/golden/ref_energy_expressions.json.  Here they are EVALUATED -- a dozen lines turn OpenMM's expression syntax
(`^`, `;`-separated definitions, `step`, `delta`) into Python -- on small random systems, summed as OpenMM sums them (every
pair once for a CustomNonbondedForce, every particle for a CustomExternalForce, every bond for a CustomBondForce), and
compared with the oracle's per-term energies in fp64.  What this pins to the reference: every energy EXPRESSION of rows a4, a5,
a7 (alternative forms), a9, a10, f1, f4 and the radii.  What stays unpinned: OpenMM's own semantics (which pairs, units of the
built-in harmonic bond / angle forces, the minimizer) -- `oracle/openmm_probe.py` is the comparison that needs OpenMM itself.
"""
import json
import math
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest

from multimm_amd.system import ChromatinSystem, ForceFieldParams, chrom_strength_per_bead, set_radiuses
from oracle.oracle import Oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = json.load(open(os.path.join(GOLD, "ref_energy_expressions.json")))
FUNCS = {"exp": math.exp, "sqrt": math.sqrt, "sin": math.sin, "cos": math.cos, "log": math.log, "abs": abs, "min": min, "max": max,
         "step": lambda v: 1.0 if v >= 0.0 else 0.0, "delta": lambda v: 1.0 if v == 0.0 else 0.0}
T_EV, T_GAUSS, T_BOND, T_ANGLE, T_LOOP, T_CONT, T_LAM, T_CENT, T_CHB = range(9)
OFF = dict(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False, EV_USE_EXCLUDED_VOLUME=False,
           NB_CUTOFF=0.0)   # every term off, NoCutoff (the reference's semantics): each test switches its own term on


def lepton(text: str, env: dict) -> float:
    """OpenMM (Lepton) expression `main; name = expr; ...` at the values of `env`."""
    def py(e):
        return re.sub(r"\blambda\b", "lambda_", e.replace("^", "**"))
    parts = [p.strip() for p in text.split(";") if p.strip()]
    ns = dict(FUNCS)
    ns.update({("lambda_" if k == "lambda" else k): float(v) for k, v in env.items()})
    pending = {k.strip(): py(v) for k, v in (p.split("=", 1) for p in parts[1:])}
    for _ in range(len(pending) + 1):
        for k, e in list(pending.items()):
            try:
                ns[k] = eval(e, {"__builtins__": {}}, ns)
                del pending[k]
            except NameError:
                pass
    assert not pending, pending
    return float(eval(py(parts[0]), {"__builtins__": {}}, ns))


def branch(fn: str, mode: str | None) -> tuple[str, dict, dict]:
    """(expression text, {global parameter: source text of its value}, {local name: source text}) of a force builder's branch"""
    entry = REF["functions"][fn]
    b = entry["branches"][mode] if mode else entry["common"]
    assert len(b["expressions"]) == 1, (fn, mode, b["expressions"])
    g = dict(entry["common"]["globals"])
    g.update(b["globals"])
    loc = dict(entry["common"]["locals"])
    loc.update(b["locals"])
    return b["expressions"][0]["text"], g, loc


def values(globals_src: dict, locals_src: dict, s: ChromatinSystem, extra: dict | None = None) -> dict:
    """the parameter values the reference would hand over: its own source text (`self.args.COB_EA`, `0.1 * (self.radius2 -
    self.radius1)`, ...) evaluated on this system's configuration, radii and mass centre"""
    R1, R2, r_comp = s.radii
    me = SimpleNamespace(args=SimpleNamespace(**{k: getattr(s.ff, k) for k in vars(s.ff)}), radius1=R1, radius2=R2, r_comp=r_comp,
                         mass_center=s.centre)
    ns = {"self": me, "np": np, "sigma_val": s.ff.LE_HARMONIC_BOND_R0}       # (add_evforce: sigma = LE_HARMONIC_BOND_R0, model.py:175-179)
    ns.update(extra or {})
    for k, src in locals_src.items():
        ns[k] = eval(src, {"__builtins__": {}}, ns)
    return {k: float(eval(src, {"__builtins__": {}}, ns)) for k, src in globals_src.items()}


def small_system(n=14, seed=0, two_chroms=True, spread=0.35, **ff):
    rng = np.random.default_rng(seed)
    x = rng.normal(0.0, spread, (n, 3))
    ends = np.array([0, n // 2, n] if two_chroms else [0, n])
    s = ChromatinSystem(n, x, ends, rng.choice(np.array([-2, -1, 0, 1, 2], np.int8), n), ff=ForceFieldParams(**{**OFF, **ff}))
    s.chrom_strength = chrom_strength_per_bead(s.chr_ends, n)
    return s


def oracle_terms(s):
    return Oracle(s, as_float32_inputs=False).eval()[0]


# ---- the same cases as (system, reference sum, term index): tests/test_gpu_reference_expressions.py runs them on the device ----
def _pairs(s, text, p, per_pair, x=None):
    x = s.positions if x is None else x
    return sum(lepton(text, {**p, "r": np.linalg.norm(x[i] - x[j]), **per_pair(i, j)}) for j in range(s.n_beads) for i in range(j))


def expression_cases():
    out = []
    for mode in ("powerlaw", "gaussian_core"):
        s = small_system(EV_USE_EXCLUDED_VOLUME=True, EV_FORCE_TYPE=mode)
        text, g, loc = branch("add_evforce", mode)
        p = values(g, loc, s)
        out.append((f"ev-{mode}", s, _pairs(s, text, p, lambda i, j: {}), T_EV,
                    lambda x, s=s, text=text, p=p: _pairs(s, text, p, lambda i, j: {}, x)))
    for fn, key, switch, tag in (("add_compartment_blocks", "COB_FORCE_TYPE", "COB_USE_COMPARTMENT_BLOCKS", "cob"),
                                 ("add_subcompartment_blocks", "SCB_FORCE_TYPE", "SCB_USE_SUBCOMPARTMENT_BLOCKS", "scb")):
        for mode in ("gaussian", "yukawa", "theta"):
            s = small_system(seed=3, spread=0.12, **{switch: True, key: mode})
            text, g, loc = branch(fn, mode)
            lab = s.labels.astype(float)
            p = values(g, loc, s)
            pp = lambda i, j, lab=lab: {"s1": lab[i], "s2": lab[j]}
            out.append((f"{tag}-{mode}", s, _pairs(s, text, p, pp), T_GAUSS, lambda x, s=s, text=text, p=p, pp=pp: _pairs(s, text, p, pp, x)))
    for mode in ("polynomial", "gaussian", "saturating"):
        s = small_system(seed=5, CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_FORCE_TYPE=mode)
        text, g, loc = branch("add_chromosomal_blocks", mode)
        chrom = np.searchsorted(s.chr_ends, np.arange(s.n_beads), side="right") - 1
        p = values(g, loc, s)
        pp = lambda i, j, chrom=chrom: {"chrom1": chrom[i], "chrom2": chrom[j]}
        out.append((f"chb-{mode}", s, _pairs(s, text, p, pp), T_CHB, lambda x, s=s, text=text, p=p, pp=pp: _pairs(s, text, p, pp, x)))
    s = small_system(n=40, seed=7, SC_USE_SPHERICAL_CONTAINER=True)
    s.positions *= 0.5 * s.radii[1] / 0.35
    text, g, loc = branch("add_spherical_container", None)
    p = values(g, loc, s)
    out.append(("container", s, _external(s, text, p), T_CONT, lambda x, s=s, text=text, p=p: _external(s, text, p, None, x)))
    for mode in ("sin", "gaussian_shell", "harmonic_shell", "logistic_shell"):
        s = small_system(n=40, seed=9, IBL_USE_B_LAMINA_INTERACTION=True, BLAMINA_FORCE_TYPE=mode)
        s.positions *= 0.5 * s.radii[1] / 0.35
        text, g, loc = branch("add_Blamina_interaction", mode)
        p, per = values(g, loc, s), {"s": s.labels.astype(float)}
        out.append((f"lamina-{mode}", s, _external(s, text, p, per), T_LAM, lambda x, s=s, text=text, p=p, per=per: _external(s, text, p, per, x)))
    for mode in ("harmonic", "gaussian", "logistic"):
        s = small_system(n=40, seed=11, CF_USE_CENTRAL_FORCE=True, CENTRAL_FORCE_TYPE=mode)
        s.positions *= 0.5 * s.radii[1] / 0.35
        text, g, loc = branch("add_central_force", mode)
        p, per = values(g, loc, s), {"chrom_s": s.chrom_strength}
        out.append((f"central-{mode}", s, _external(s, text, p, per), T_CENT, lambda x, s=s, text=text, p=p, per=per: _external(s, text, p, per, x)))
    return out


def test_case_table_agrees_with_the_oracle():
    for name, s, want, term, energy_of in expression_cases():
        assert oracle_terms(s)[term] == pytest.approx(want, rel=1e-12), name
        assert energy_of(s.positions) == want


def test_radii_follow_the_reference_text():
    for n in (1000, 50000, 200000, 1000000):
        ns = {"b0": 0.1, "N": float(n)}
        for a in REF["set_radiuses"]:
            ns[a["name"]] = eval(a["value"], {"__builtins__": {}}, ns)
        assert set_radiuses(n, 0.1) == pytest.approx((ns["R1"], ns["R2"], ns["r_comp"]), rel=1e-15)


@pytest.mark.parametrize("mode", ["powerlaw", "gaussian_core"])
def test_excluded_volume_expression(mode):
    s = small_system(EV_USE_EXCLUDED_VOLUME=True, EV_FORCE_TYPE=mode)
    text, g, loc = branch("add_evforce", mode)
    p = values(g, loc, s)
    x = s.positions
    want = sum(lepton(text, {**p, "r": np.linalg.norm(x[i] - x[j])}) for i in range(s.n_beads) for j in range(i))
    assert abs(want) > 1.0 and oracle_terms(s)[T_EV] == pytest.approx(want, rel=1e-12)


@pytest.mark.parametrize("fn, key, switch", [("add_compartment_blocks", "COB_FORCE_TYPE", "COB_USE_COMPARTMENT_BLOCKS"),
                                             ("add_subcompartment_blocks", "SCB_FORCE_TYPE", "SCB_USE_SUBCOMPARTMENT_BLOCKS")])
@pytest.mark.parametrize("mode", ["gaussian", "yukawa", "theta"])
def test_compartment_expressions(fn, key, switch, mode):
    """COB yukawa reads `s1` twice (model.py:262-267): the text is evaluated as written, with particle 1 = the lower index of a
    pair, which is the convention the oracle documents for that quirk."""
    s = small_system(seed=3, spread=0.12, **{switch: True, key: mode})    # (r_comp = 0.15 nm: pairs on both sides of the theta form's step)
    text, g, loc = branch(fn, mode)
    p = values(g, loc, s)
    x, lab = s.positions, s.labels.astype(float)
    want = sum(lepton(text, {**p, "r": np.linalg.norm(x[i] - x[j]), "s1": lab[i], "s2": lab[j]})
               for j in range(s.n_beads) for i in range(j))
    assert abs(want) > 0.1 and oracle_terms(s)[T_GAUSS] == pytest.approx(want, rel=1e-12)


@pytest.mark.parametrize("mode", ["polynomial", "gaussian", "saturating"])
def test_chromosomal_block_expressions(mode):
    s = small_system(seed=5, CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_FORCE_TYPE=mode)
    text, g, loc = branch("add_chromosomal_blocks", mode)
    p = values(g, loc, s)
    x = s.positions
    chrom = np.searchsorted(s.chr_ends, np.arange(s.n_beads), side="right") - 1
    want = sum(lepton(text, {**p, "r": np.linalg.norm(x[i] - x[j]), "chrom1": chrom[i], "chrom2": chrom[j]})
               for j in range(s.n_beads) for i in range(j))
    assert abs(want) > 1e-5 and oracle_terms(s)[T_CHB] == pytest.approx(want, rel=1e-12)


def _external(s, text, p, per_particle=None, x=None):
    c = s.centre            # (a global parameter of the force: set once, from the structure the system was built with)
    pos = s.positions if x is None else x
    tot = 0.0
    for i in range(s.n_beads):
        env = {**p, "x": pos[i, 0], "y": pos[i, 1], "z": pos[i, 2], "x0": c[0], "y0": c[1], "z0": c[2]}
        if per_particle:
            env.update({k: v[i] for k, v in per_particle.items()})
        tot += lepton(text, env)
    return tot


def test_container_expression():
    s = small_system(n=40, seed=7, SC_USE_SPHERICAL_CONTAINER=True)
    s.positions *= 0.5 * s.radii[1] / 0.35          # beads inside the core, in the shell and outside the container
    text, g, loc = branch("add_spherical_container", None)
    want = _external(s, text, values(g, loc, s))
    assert abs(want) > 1.0 and oracle_terms(s)[T_CONT] == pytest.approx(want, rel=1e-12)


@pytest.mark.parametrize("mode", ["sin", "gaussian_shell", "harmonic_shell", "logistic_shell"])
def test_lamina_expressions(mode):
    s = small_system(n=40, seed=9, IBL_USE_B_LAMINA_INTERACTION=True, BLAMINA_FORCE_TYPE=mode)
    s.positions *= 0.5 * s.radii[1] / 0.35
    text, g, loc = branch("add_Blamina_interaction", mode)
    want = _external(s, text, values(g, loc, s), {"s": s.labels.astype(float)})
    assert abs(want) > 1.0 and oracle_terms(s)[T_LAM] == pytest.approx(want, rel=1e-12)


@pytest.mark.parametrize("mode", ["harmonic", "gaussian", "logistic"])
def test_central_force_expressions(mode):
    s = small_system(n=40, seed=11, CF_USE_CENTRAL_FORCE=True, CENTRAL_FORCE_TYPE=mode)
    s.positions *= 0.5 * s.radii[1] / 0.35
    text, g, loc = branch("add_central_force", mode)
    want = _external(s, text, values(g, loc, s), {"chrom_s": s.chrom_strength})
    assert abs(want) > 1e-3 and oracle_terms(s)[T_CENT] == pytest.approx(want, rel=1e-12)


@pytest.mark.parametrize("mode", ["fene_soft", "gaussian_tether"])
def test_loop_expressions(mode):
    """Per-bond parameters as `add_loops` computes them for every loop (model.py:675-680, 696-701): r0 = ds[i], k =
    LE_HARMONIC_BOND_K, alpha = 1 / r0^2 or sigma = r0 / 2 -- the latter two from the reference's own text."""
    rng = np.random.default_rng(13)
    s = small_system(n=30, seed=13, two_chroms=False, LE_USE_HARMONIC_BOND=True, LE_LOOP_FORCE_TYPE=mode)
    m = rng.integers(0, 14, 8)
    n = m + rng.integers(3, 15, 8)
    s.loop_m, s.loop_n, s.loop_r0 = m.astype(np.int32), n.astype(np.int32), rng.uniform(0.1, 0.2, 8)
    text, g, loc = branch("add_loops", mode)
    want = 0.0
    for a, b, r0 in zip(s.loop_m, s.loop_n, s.loop_r0):
        ns = {"r0": float(r0)}
        extra = {k: float(eval(src, {"__builtins__": {}}, ns)) for k, src in loc.items() if k not in ("r0", "k")}
        want += lepton(text, {"r": np.linalg.norm(s.positions[a] - s.positions[b]), "r0": r0, "k": s.ff.LE_HARMONIC_BOND_K, **extra})
    assert abs(want) > 1.0 and oracle_terms(s)[T_LOOP] == pytest.approx(want, rel=1e-12)


def test_call_sites_that_fix_the_semantics():
    """`simulation.minimizeEnergy()` is called bare (model.py:886): OpenMM's defaults, tolerance 10 kJ/mol/nm and no iteration
    limit -- this engine's MIN_TOLERANCE / MIN_MAX_ITERATIONS defaults; no force of the reference is ever given a nonbonded
    method, a cutoff, a switching function or an exclusion: every pair interacts (OpenMM's NoCutoff default), bonded neighbours
    included -- what `NB_CUTOFF <= 0` reproduces and what the pair sums above assume."""
    from multimm_amd.config import SimulationConfig
    calls = REF["calls"]
    assert len(calls["minimizeEnergy"]) == 1 and calls["minimizeEnergy"][0]["args"] == [] and calls["minimizeEnergy"][0]["keywords"] == {}
    assert (SimulationConfig().MIN_TOLERANCE, SimulationConfig().MIN_MAX_ITERATIONS) == (10.0, 0)
    for k in ("setNonbondedMethod", "setCutoffDistance", "setUseSwitchingFunction", "addExclusion", "createExclusionsFromBonds"):
        assert calls[k] == 0, k


@pytest.mark.parametrize("ends", [[0, 40], [0, 7, 19, 40], [0, 1, 2, 40], [0, 38, 39, 40]])
def test_backbone_topology_follows_the_reference_conditions(ends):
    """Which beads get a bond / an angle: the reference's loop range and `if` condition (model.py:625-636, 708-720, read as text:
    `i not in self.chr_ends` -- chr_ends holds the FIRST bead of every chromosome and N, hence the off-by-one quirks of SURVEY.md
    appendix A.2) evaluated for every i, against `system.backbone_flags`; the arguments of addBond / addAngle name beads
    i, i + 1 (, i + 2) and the POL_* keys this engine reads."""
    from multimm_amd.system import backbone_flags
    n = ends[-1]
    flags = backbone_flags(n, np.array(ends))
    me = SimpleNamespace(chr_ends=np.array(ends), system=SimpleNamespace(getNumParticles=lambda: n))
    for fn, bit in (("add_harmonic_bonds", 1), ("add_stiffness", 2)):
        b = REF["backbone"][fn]
        upper = eval(b["range"], {"__builtins__": {}}, {"self": me})
        want = np.zeros(n, bool)
        for i in range(upper):
            want[i] = bool(eval(b["condition"], {"__builtins__": {}}, {"self": me, b["index"]: i}))
        assert np.array_equal((flags & bit) != 0, want), (fn, ends)
    assert REF["backbone"]["add_harmonic_bonds"]["args"] == ["i", "i + 1", "self.args.POL_HARMONIC_BOND_R0", "self.args.POL_HARMONIC_BOND_K"]
    assert REF["backbone"]["add_stiffness"]["args"] == ["i", "i + 1", "i + 2", "self.args.POL_HARMONIC_ANGLE_R0",
                                                       "self.args.POL_HARMONIC_ANGLE_CONSTANT_K"]
    loops = REF["functions"]["add_loops"]["branches"]["harmonic"]["locals"]
    assert loops == {"r0": "self.args.LE_HARMONIC_BOND_R0 if self.args.LE_FIXED_DISTANCES else self.ds[i]",
                     "k": "self.args.LE_HARMONIC_BOND_K"}       # what ChromatinSystem.loop_rest_lengths() / the loop kernel use


def test_integrator_wiring_follows_the_reference():
    """SIM_INTEGRATOR_TYPE -> OpenMM integrator and the keys it is built from, in argument order (model.py:768-808, read as
    text): the four fixed-step ones are what this engine provides, fed from the same keys (`MultiMM.run_md`); the two
    variable-step ones read `self.SIM_ERROR_TOLERANCE` -- an attribute the class never sets (the key lives in `self.args`) --
    so the reference itself cannot construct them.  Velocities: setVelocitiesToTemperature(SIM_TEMPERATURE,
    SHUFFLING_SEED) (model.py:878).  Bead mass: the one atom type of forcefields/ff.xml."""
    import inspect
    from multimm_amd import engine, model
    ints = REF["integrators"]
    fixed = {k for k, v in ints.items() if not any("SIM_ERROR_TOLERANCE" in a for a in v["args"])}
    assert fixed == set(engine.INTEGRATORS) == {"langevin", "verlet", "brownian", "amd"}
    assert ints["langevin"]["args"] == ints["brownian"]["args"] == [
        "self.args.SIM_TEMPERATURE", "self.args.SIM_FRICTION_COEFF", "self.args.SIM_INTEGRATOR_STEP"]
    assert ints["verlet"]["args"] == ["self.args.SIM_INTEGRATOR_STEP"]
    assert ints["amd"]["args"] == ["self.args.SIM_INTEGRATOR_STEP", "self.args.SIM_AMD_ALPHA", "self.args.SIM_AMD_E"]
    for k in ("variable_verlet", "variable_langevin"):
        assert "self.SIM_ERROR_TOLERANCE" in ints[k]["args"] and not any("self.args.SIM_ERROR_TOLERANCE" in a for a in ints[k]["args"])
    assert REF["set_velocities"] == [{"args": ["self.args.SIM_TEMPERATURE", "self.args.SHUFFLING_SEED"], "line": REF["set_velocities"][0]["line"]}]
    src = inspect.getsource(model.MultiMM.run_md)
    for key in ("SIM_INTEGRATOR_STEP", "SIM_TEMPERATURE", "SIM_FRICTION_COEFF", "SIM_AMD_ALPHA", "SIM_AMD_E", "SHUFFLING_SEED"):
        assert key in src, key
    assert REF["atom_type_masses"] == [engine.BEAD_MASS_AMU]


def test_every_switch_gates_the_term_the_reference_builds_with_it():
    """`add_forcefield` (model.py:812-857, read as text): ten `*_USE_*` switches, each guarding one force builder.  With exactly one
    switch on, exactly the energy term of that builder is non-zero here (oracle), all nine others vanish."""
    builder_term = {"add_evforce": T_EV, "add_compartment_blocks": T_GAUSS, "add_subcompartment_blocks": T_GAUSS,
                    "add_chromosomal_blocks": T_CHB, "add_spherical_container": T_CONT, "add_Blamina_interaction": T_LAM,
                    "add_central_force": T_CENT, "add_harmonic_bonds": T_BOND, "add_loops": T_LOOP, "add_stiffness": T_ANGLE}
    table = REF["add_forcefield"]
    assert [t["builder"] for t in table] == list(builder_term)          # source order: the reference's (and this table's)
    rng = np.random.default_rng(21)
    for t in table:
        key = t["switch"].removeprefix("self.args.")
        assert key in vars(ForceFieldParams()), key
        s = small_system(n=40, seed=23, spread=0.12, **{key: True})
        s.positions *= 3.0 if key in ("SC_USE_SPHERICAL_CONTAINER", "IBL_USE_B_LAMINA_INTERACTION", "CF_USE_CENTRAL_FORCE") else 1.0
        s.loop_m, s.loop_n = np.array([1, 5], np.int32), np.array([9, 30], np.int32)
        s.loop_r0 = np.array([0.1, 0.15])
        et = oracle_terms(s)
        term = builder_term[t["builder"]]
        assert abs(et[term]) > 1e-9, (key, et)
        assert np.abs(np.delete(et, term)).max() == 0.0, (key, et)
