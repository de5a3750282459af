"""Defaults table of the reference's ini schema as DATA: the fields of `SimulationConfig` (`/root/reference/src/multimm/config.py:94-312`)
read from the source TEXT with `ast` -- nothing of the reference is imported or executed (its module needs OpenMM at import) --
name, annotation and the literal `default=` of every `Field(...)`.  Build container only; the output,
tests/golden/ref_config_defaults.json, is what tests/test_reference_fixtures.py compares `multimm_amd.config` with.
usage: python scripts/make_reference_config_fixture.py [/root/reference]"""
import ast, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
src_path = os.path.join(root, "src", "multimm", "config.py")
tree = ast.parse(open(src_path).read())
fields = []
for node in tree.body:
    if isinstance(node, ast.ClassDef) and node.name == "SimulationConfig":
        for st in node.body:
            if not (isinstance(st, ast.AnnAssign) and isinstance(st.target, ast.Name) and isinstance(st.value, ast.Call)):
                continue
            if getattr(st.value.func, "id", None) != "Field":
                continue
            entry = {"name": st.target.id, "annotation": ast.unparse(st.annotation), "line": st.lineno}
            for kw in st.value.keywords:
                if kw.arg == "default":
                    try:
                        entry["default"] = ast.literal_eval(kw.value)
                        entry["literal"] = True
                    except ValueError:
                        entry["default"] = ast.unparse(kw.value)      # a name or an enum member: kept as source text
                        entry["literal"] = False
            fields.append(entry)
# the values INITIAL_STRUCTURE_TYPE may take: members of enums.py:InitialStructureType, read the same way
enum_tree = ast.parse(open(os.path.join(root, "src", "multimm", "enums.py")).read())
kinds = [ast.literal_eval(st.value) for node in enum_tree.body if isinstance(node, ast.ClassDef) and node.name == "InitialStructureType"
         for st in node.body if isinstance(st, ast.Assign)]
# the names each *_FORCE_TYPE key may take: in model.py every force builder reads `mode = getattr(self.args, "<KEY>", "<default>")`
# and branches on `mode == "<name>"` (model.py:173-215, 229-292, 305-382, 395-449, 479-544, 557-615, 648-...)
forms = {}
for fn in ast.walk(ast.parse(open(os.path.join(root, "src", "multimm", "model.py")).read())):
    if not isinstance(fn, ast.FunctionDef):
        continue
    key = default = None
    for n in ast.walk(fn):
        if (isinstance(n, ast.Assign) and isinstance(n.value, ast.Call) and getattr(n.value.func, "id", "") == "getattr"
                and len(n.value.args) == 3 and isinstance(n.value.args[1], ast.Constant)
                and str(n.value.args[1].value).endswith("FORCE_TYPE") and getattr(n.targets[0], "id", "") == "mode"):
            key, default = n.value.args[1].value, ast.literal_eval(n.value.args[2])
    if key is None:
        continue
    names = []
    for n in ast.walk(fn):
        if isinstance(n, ast.Compare) and getattr(n.left, "id", "") == "mode":
            for c in n.comparators:
                if isinstance(c, ast.Constant) and isinstance(c.value, str) and c.value not in names:
                    names.append(c.value)
    forms[key] = {"default": default, "names": names, "function": fn.name, "line": fn.lineno}
# the MODELLING_LEVEL presets: run.py's ArgumentChanger.convenient_argument_changer is an if / elif chain on `level` whose branches
# call self.set_arg("<KEY>", <value>); values are literals, bool(self.args.COMPARTMENT_PATH) ("has_compartments") or expressions
# on the chromosome sizes (kept as source text)
levels = []
for fn in ast.walk(ast.parse(open(os.path.join(root, "src", "multimm", "run.py")).read())):
    if isinstance(fn, ast.FunctionDef) and fn.name == "convenient_argument_changer":
        def names_of(test):
            if isinstance(test.comparators[0], ast.Constant):
                return [test.comparators[0].value]
            return [e.value for e in test.comparators[0].elts]
        def sets_of(body):
            out_ = {}
            for st in body:
                call = getattr(st, "value", None)
                if isinstance(call, ast.Call) and getattr(call.func, "attr", "") == "set_arg":
                    key = call.args[0].value
                    try:
                        out_[key] = ast.literal_eval(call.args[1])
                    except ValueError:
                        src = ast.unparse(call.args[1])
                        out_[key] = "has_compartments" if src == "bool(self.args.COMPARTMENT_PATH)" else {"expr": src}
            return out_
        always = sets_of(fn.body)
        node = next(st for st in fn.body if isinstance(st, ast.If))
        while node is not None:
            levels.append({"names": names_of(node.test), "sets": sets_of(node.body), "line": node.lineno})
            nxt = node.orelse
            node = nxt[0] if len(nxt) == 1 and isinstance(nxt[0], ast.If) else None
out = {"source": "src/multimm/config.py (SimulationConfig), read as text with ast; line = line of the field", "fields": fields,
       "initial_structure_types": kinds, "force_types": forms, "modelling_levels": levels, "modelling_levels_always": always}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_config_defaults.json")
json.dump(out, open(dst, "w"), indent=1)
print(len(fields), "fields ->", dst)
