"""Builds libmmx.so (HIP, gfx950) in-tree with hipcc.  `python -m multimm_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libmmx.so")
SRC = os.path.join(HERE, "csrc", "mmx_api.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))] + [
    os.path.join(ROOT, "include", "mmx.h")
]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmmx.so cannot be built (ROCm toolchain required)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into multimm_amd/libmmx.so; returns the path."""
    if not force and not needs_build():
        return LIB
    cmd = [
        hipcc_path(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
        "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-unused-const-variable",
        "-o", LIB + ".tmp", SRC,
    ]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
