"""The C-ABI library loads and exports every symbol include/mmx.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mmx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmx_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from multimm_amd import build
    from multimm_amd.engine import SIGNATURES, load_library
    build.build()
    lib = load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mmx.h but not exported by libmmx.so"
        assert n in SIGNATURES, f"{n} declared in mmx.h but not bound in multimm_amd/engine.py"
    assert set(SIGNATURES) == set(names)
    assert lib.mmx_abi_version() == 1


def test_stats_struct_layout_matches_header():
    from multimm_amd.engine import MMXStats
    # 4 int32 + 6 double + 9 double (terms) + 8 double + 8 int64 + 8 int64 (kernel slots)
    assert ctypes.sizeof(MMXStats) == 16 + 8 * (6 + 9 + 8 + 8 + 8)


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from multimm_amd.engine import Engine, MMXError
    with pytest.raises(MMXError) as ei:
        Engine(100)
    assert ei.value.code == -2


def test_product_package_never_imports_oracle():
    import ast
    pkg = os.path.join(ROOT, "multimm_amd")
    for f in os.listdir(pkg):
        if not f.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, f)).read())
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""]
            assert not any(m.split(".")[0] == "oracle" for m in mods), f"{f} imports the oracle"
    for f in os.listdir(os.path.join(pkg, "csrc")):
        assert "oracle" not in open(os.path.join(pkg, "csrc", f)).read().lower(), f
