"""The reference's energy expressions as DATA: every string its force builders hand to OpenMM's Custom*Force -- the constructor
argument or `setEnergyFunction(...)` of `model.py`'s `add_*` methods, per `*_FORCE_TYPE` branch -- with the names of the global /
per-particle / per-bond parameters and the source text of what they are set to, read from the source TEXT with `ast`
(`/root/reference/src/multimm/model.py:164-720`; nothing is imported or executed: the module needs OpenMM).  Build container
only.  Output: tests/golden/ref_energy_expressions.json, which tests/test_reference_expressions.py evaluates numerically against
the fp64 oracle's energies.      usage: python scripts/make_reference_energy_fixture.py [/root/reference]"""
import ast, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
tree = ast.parse(open(os.path.join(root, "src", "multimm", "model.py")).read())
PARAM_CALLS = {"addGlobalParameter": "globals", "addPerParticleParameter": "per_particle", "addPerBondParameter": "per_bond"}


def collect(nodes, strings):
    """expression strings, parameters and plain local assignments among `nodes` (statement lists are walked in full)"""
    def resolve(n):
        if isinstance(n, ast.Constant) and isinstance(n.value, str):
            return n.value
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Add):
            a, b = resolve(n.left), resolve(n.right)
            return None if a is None or b is None else a + b
        if isinstance(n, ast.Name):
            return strings.get(n.id)
        if isinstance(n, ast.JoinedStr):
            parts = [p.value if isinstance(p, ast.Constant) else resolve(p.value) for p in n.values]
            return None if any(p is None for p in parts) else "".join(parts)
        return None
    out = {"expressions": [], "globals": {}, "per_particle": [], "per_bond": [], "locals": {}}
    for st in nodes:
        for n in ast.walk(st):
            if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute):
                name = n.func.attr
                if (name == "setEnergyFunction" or name.startswith("Custom")) and n.args:
                    s = resolve(n.args[0])
                    if s is not None and s.strip() != "0":
                        out["expressions"].append({"call": name, "text": s, "line": n.lineno})
                elif name in PARAM_CALLS and n.args and isinstance(n.args[0], ast.Constant):
                    if name == "addGlobalParameter":
                        val = n.args[1] if len(n.args) > 1 else next((kw.value for kw in n.keywords if kw.arg == "defaultValue"), None)
                        out["globals"][n.args[0].value] = ast.unparse(val) if val is not None else None
                    else:
                        out[PARAM_CALLS[name]].append(n.args[0].value)
            # plain arithmetic on the radii etc. (sigma = 0.1 * (self.radius2 - self.radius1)): kept as source text
            if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) \
                    and isinstance(n.value, (ast.BinOp, ast.Constant, ast.IfExp, ast.Attribute, ast.Subscript)) and resolve(n.value) is None:
                out["locals"][n.targets[0].id] = ast.unparse(n.value)
    return out


result = {}
for fn in ast.walk(tree):
    if not (isinstance(fn, ast.FunctionDef) and fn.name.startswith("add_")):
        continue
    strings = {}
    for n in ast.walk(fn):
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) \
                and isinstance(n.value, ast.Constant) and isinstance(n.value.value, str):
            strings[n.targets[0].id] = n.value.value
    chain = None
    for st in fn.body:
        if isinstance(st, ast.If) and isinstance(st.test, ast.Compare) and getattr(st.test.left, "id", "") == "mode":
            chain = st
    entry = {"line": fn.lineno, "branches": {}}
    common_nodes = [st for st in fn.body if st is not chain]
    entry["common"] = collect(common_nodes, strings)
    node = chain
    while node is not None:
        mode = node.test.comparators[0].value
        entry["branches"][mode] = collect(node.body, strings)
        nxt = node.orelse
        node = nxt[0] if len(nxt) == 1 and isinstance(nxt[0], ast.If) and getattr(nxt[0].test.left, "id", "") == "mode" else None
    if entry["common"]["expressions"] or any(b["expressions"] for b in entry["branches"].values()):
        result[fn.name] = entry
# set_radiuses (model.py:1016-1067): its arithmetic assignments in source order (R2, inner_volume_fraction, R1, r_comp from b0, N)
radii = []
for fn in ast.walk(tree):
    if isinstance(fn, ast.FunctionDef) and fn.name == "set_radiuses":
        for st in fn.body:
            if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name) and isinstance(st.value, (ast.BinOp, ast.Constant)):
                radii.append({"name": st.targets[0].id, "value": ast.unparse(st.value), "line": st.lineno})
# call sites that fix the semantics of the path: minimizeEnergy() is called without arguments (OpenMM's defaults: tolerance
# 10 kJ/mol/nm, maxIterations 0 = until converged), and no force is ever given a nonbonded method or a cutoff (OpenMM's default
# for a CustomNonbondedForce: NoCutoff -- every pair)
calls = {"minimizeEnergy": [], "setNonbondedMethod": 0, "setCutoffDistance": 0, "setUseSwitchingFunction": 0, "addExclusion": 0,
         "createExclusionsFromBonds": 0}
for n in ast.walk(tree):
    if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute):
        if n.func.attr == "minimizeEnergy":
            calls["minimizeEnergy"].append({"line": n.lineno, "args": [ast.unparse(a) for a in n.args],
                                            "keywords": {k.arg: ast.unparse(k.value) for k in n.keywords}})
        elif n.func.attr in calls:
            calls[n.func.attr] += 1
# the backbone: which bonds and angles the built-in harmonic forces receive (model.py:625-636, 708-720): the range of the loop, the
# condition under which bead i gets a term, and the arguments of addBond / addAngle, all as source text
backbone = {}
for fn in ast.walk(tree):
    if isinstance(fn, ast.FunctionDef) and fn.name in ("add_harmonic_bonds", "add_stiffness"):
        loop = next(n for n in ast.walk(fn) if isinstance(n, ast.For))
        cond = next(n for n in ast.walk(loop) if isinstance(n, ast.If))
        call = next(n for n in ast.walk(cond) if isinstance(n, ast.Call) and getattr(n.func, "attr", "") in ("addBond", "addAngle"))
        backbone[fn.name] = {"line": fn.lineno, "range": ast.unparse(loop.iter.args[0]), "index": loop.target.id,
                             "condition": ast.unparse(cond.test), "call": call.func.attr, "args": [ast.unparse(a) for a in call.args]}
# MD: which OpenMM integrator every SIM_INTEGRATOR_TYPE constructs and from which keys, in argument order (model.py:768-808),
# and how velocities are drawn (model.py:878)
integrators = {}
for n in ast.walk(tree):
    if isinstance(n, ast.Match) and ast.unparse(n.subject) == "self.args.SIM_INTEGRATOR_TYPE":
        for case in n.cases:
            call = next(c for c in ast.walk(case) if isinstance(c, ast.Call) and ast.unparse(c.func).startswith("mm."))
            integrators[case.pattern.value.value] = {"class": ast.unparse(call.func), "args": [ast.unparse(a) for a in call.args],
                                                     "line": call.lineno}
velocities = [{"args": [ast.unparse(a) for a in n.args], "line": n.lineno} for n in ast.walk(tree)
              if isinstance(n, ast.Call) and getattr(n.func, "attr", "") == "setVelocitiesToTemperature"]
# add_forcefield (model.py:812-857): which switch gates which force builder, in source order
switches = []
for fn in ast.walk(tree):
    if isinstance(fn, ast.FunctionDef) and fn.name == "add_forcefield":
        for st in fn.body:
            if isinstance(st, ast.If):
                call = next(c for c in ast.walk(st) if isinstance(c, ast.Call) and ast.unparse(c.func).startswith("self.add_"))
                switches.append({"switch": ast.unparse(st.test), "builder": call.func.attr, "line": st.lineno})
# the one atom type of forcefields/ff.xml: the bead mass the integrators see
import re
masses = [float(m) for m in re.findall(r'<Type[^>]*mass="([^"]+)"', open(os.path.join(root, "src", "multimm", "forcefields", "ff.xml")).read())]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_energy_expressions.json")
json.dump({"source": "src/multimm/model.py, add_* methods and set_radiuses, read as text with ast", "functions": result,
           "set_radiuses": radii, "calls": calls, "backbone": backbone, "integrators": integrators,
           "set_velocities": velocities, "atom_type_masses": masses, "add_forcefield": switches}, open(dst, "w"), indent=1)
print("calls:", calls)
print("backbone:", backbone)
print("set_radiuses:", radii)
for k, v in result.items():
    for m, b in list(v["branches"].items()) + [("(no branch)", v["common"])]:
        for e in b["expressions"]:
            print(f"{k:28s} {m:16s} {e['text'][:110]}")
