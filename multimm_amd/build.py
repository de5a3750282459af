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


STAMP = LIB + ".stamp"


def source_stamp() -> str:
    """Hash of everything the library is built from (csrc/, include/mmx.h, this recipe) plus the compiler's version:
    the library on disk is current iff the stamp written beside it matches -- file times do not survive a copy to
    another machine, content does."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(DEPS) + [os.path.abspath(__file__)]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    try:
        ver = subprocess.run([hipcc_path(), "--version"], capture_output=True, text=True).stdout
    except Exception:
        ver = "no hipcc"
    h.update(ver.encode())
    return h.hexdigest()


def needs_build() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    try:
        return open(STAMP).read().strip() != source_stamp()
    except OSError:
        return True


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
    with open(STAMP, "w") as f:
        f.write(source_stamp() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
