"""Hilbert-curve start structure.

Reference: ``initial_structure_tools.py:157-166`` calls ``HilbertCurve(p=8, n=3).points_from_distances``
of the third-party package hilbertcurve 2.0.5 (``uv.lock:1129-1130``; Skilling, "Programming the
Hilbert curve", 2004).  This is a vectorised numpy restatement of that published algorithm
(integer -> transpose -> Gray decode -> undo excess work).  The integer lattice is written to mmCIF
in Angstrom and read back by OpenMM's PDBxFile as nm (x0.1): lattice spacing 0.1 nm == bond r0.
"""
from __future__ import annotations

import numpy as np


def hilbert_points(n_points: int, p: int = 8, n: int = 3) -> np.ndarray:
    """First ``n_points`` points of the order-``p`` Hilbert curve in ``n`` dimensions, int32 [n_points, n]."""
    if n_points < 0 or n_points > (1 << (p * n)):
        raise ValueError(f"n_points={n_points} outside [0, 2^{p * n}]")
    h = np.arange(n_points, dtype=np.int64)
    x = [np.zeros(n_points, dtype=np.int64) for _ in range(n)]
    nbits = p * n
    for b in range(nbits):  # MSB-first bit string; coordinate i takes bits i, i+n, i+2n, ...
        bit = (h >> (nbits - 1 - b)) & 1
        x[b % n] = (x[b % n] << 1) | bit
    # Gray decode by H ^ (H/2)
    t = x[n - 1] >> 1
    for i in range(n - 1, 0, -1):
        x[i] = x[i] ^ x[i - 1]
    x[0] = x[0] ^ t
    # undo excess work
    q = 2
    z = 2 << (p - 1)
    while q != z:
        pm = q - 1
        for i in range(n - 1, -1, -1):
            hit = (x[i] & q) != 0
            # where hit: invert low bits of x[0]; else exchange low bits of x[0] and x[i]
            tt = (x[0] ^ x[i]) & pm
            x0_new = np.where(hit, x[0] ^ pm, x[0] ^ tt)
            xi_new = np.where(hit, x[i], x[i] ^ tt)
            if i == 0:
                x[0] = np.where(hit, x[0] ^ pm, x[0])
            else:
                x[0] = x0_new
                x[i] = xi_new
        q <<= 1
    return np.stack(x, axis=1).astype(np.int32)
