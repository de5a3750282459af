"""Initial structures of ``INITIAL_STRUCTURE_TYPE`` (config.py:138-141): what ``compute_init_struct``
(initial_structure_tools.py:256-289) hands to ``build_init_mmcif``, in the file's length unit (the mmCIF stores the
numbers as Angstrom with three decimals; OpenMM reads them back as nm / 10, model.py:733-750).

The deterministic curves (hilbert, circle, helix, spiral, knot) are the reference's formulas.  The random ones (rw,
confined_rw, self_avoiding_rw, sphere) draw from ``numpy.random.RandomState(seed)`` -- the legacy generator the
reference's global ``np.random`` calls use -- in the reference's order of draws, but the reference leaves the stream
wherever the parsers left it, so equality with a reference run is statistical, not bead by bead.
"""
from __future__ import annotations

import numpy as np

from .hilbert import hilbert_points

MODES = ("rw", "confined_rw", "knot", "self_avoiding_rw", "circle", "helix", "spiral", "sphere", "hilbert")


def _unit_rows(v: np.ndarray) -> np.ndarray:
    nrm = np.linalg.norm(v, axis=1, keepdims=True)
    out = np.divide(v, nrm, out=np.zeros_like(v), where=nrm > 0)
    out[nrm[:, 0] == 0] = (1.0, 0.0, 0.0)
    return out


def random_walk(n: int, rng, step: float = 1.0) -> np.ndarray:
    """initial_structure_tools.py:240-253: unit steps in uniformly random directions from the origin."""
    pts = np.zeros((n, 3))
    if n > 1:
        pts[1:] = np.cumsum(step * _unit_rows(rng.normal(size=(n - 1, 3))), axis=0)
    return pts


def confined_random_walk(n: int, rng, box: float = 5.0) -> np.ndarray:
    """initial_structure_tools.py:220-227: +-1 per axis and step, coordinates clipped to [-box, box] as they go."""
    steps = rng.choice([-1, 1], size=(max(n - 1, 0), 3)).astype(np.float64)
    pts = np.zeros((n, 3))
    cur = np.zeros(3)
    for i in range(1, n):
        cur = np.clip(cur + steps[i - 1], -box, box)
        pts[i] = cur
    return pts


def self_avoiding_random_walk(n: int, rng, step: float = 1.0, bead_radius: float = 0.5, eps: float = 1e-3,
                              max_trials: int = 1000) -> np.ndarray:
    """initial_structure_tools.py:614-640: a unit step is redrawn while it lands within 2 r - eps of any earlier bead,
    at most ``max_trials`` rejections per walk position altogether (then the last draw is kept).  O(n^2) like the
    reference: meant for small systems."""
    pts = np.zeros((n, 3))
    dmin = 2.0 * bead_radius - eps
    for i in range(1, n):
        trials = 0
        while True:
            v = rng.normal(0.0, 1.0, 3)
            nv = np.linalg.norm(v)
            cand = pts[i - 1] + step * (v / nv if nv > 0 else np.array([1.0, 0.0, 0.0]))
            if not np.any(np.linalg.norm(pts[:i] - cand, axis=1) < dmin):
                break
            trials += 1
            if trials >= max_trials:  # the reference gives up and keeps the last (clashing) draw
                break
        pts[i] = cand
    return pts


def circle(n: int, z_stretch: float = 50.0, radius: float = 5.0) -> np.ndarray:
    """polymer_circle(n, 50, 5), initial_structure_tools.py:169-182, 268-269: one turn of a helix of radius 5 that
    climbs z_stretch in total.  The reference's order of operations is kept (angle in degrees, ``inc * i * pi / 180``
    left to right; z accumulated step by step), so the float64 values are bit-equal
    (tests/test_reference_fixtures.py)."""
    inc = 360 / float(n)
    ang = inc * np.arange(n) * np.pi / 180
    z = np.cumsum(np.full(n, z_stretch / n))          # sequential additions, as the reference's ``z += z_stretch``
    return np.stack([radius * np.cos(ang), radius * np.sin(ang), z], axis=1)


def helix(n: int, radius: float = 1.0, pitch: float = 2.0) -> np.ndarray:
    """initial_structure_tools.py:185-191: two turns, z from 0 to pitch * n."""
    th = np.linspace(0.0, 4.0 * np.pi, n)
    return np.column_stack((radius * np.cos(th), radius * np.sin(th), np.linspace(0.0, pitch * n, n)))


def spiral(n: int, initial_radius: float = 1.0, pitch: float = 1.0, growth: float = 0.05) -> np.ndarray:
    """initial_structure_tools.py:194-201: two turns with the radius growing by ``growth`` per bead."""
    th = np.linspace(0.0, 4.0 * np.pi, n)
    r = initial_radius + growth * np.arange(n)
    return np.column_stack((r * np.cos(th), r * np.sin(th), np.linspace(0.0, pitch * n, n)))


def sphere(n: int, rng, radius: float = 1.0) -> np.ndarray:
    """initial_structure_tools.py:204-217: despite its name, points uniform in the BALL of ``radius`` (r = R u^(1/3))."""
    phi = rng.uniform(0.0, 2.0 * np.pi, n)
    cost = rng.uniform(-1.0, 1.0, n)
    u = rng.uniform(0.0, 1.0, n)
    th = np.arccos(cost)
    r = radius * u ** (1.0 / 3.0)
    return np.column_stack((r * np.sin(th) * np.cos(phi), r * np.sin(th) * np.sin(phi), r * np.cos(th)))


def trefoil_knot(n: int, scale: float = 5.0) -> np.ndarray:
    """initial_structure_tools.py:230-237."""
    t = np.linspace(0.0, 2.0 * np.pi, n)
    return np.column_stack((scale * (np.sin(t) + 2.0 * np.sin(2.0 * t)), scale * (np.cos(t) - 2.0 * np.cos(2.0 * t)),
                            -scale * np.sin(3.0 * t)))


def compute_init_struct(n_beads: int, mode: str = "hilbert", seed: int = 0) -> np.ndarray:
    """[n_beads, 3] float64 in the mmCIF's unit (initial_structure_tools.py:256-289; same error text)."""
    n = int(n_beads)
    mode = str(mode).lower()
    rng = np.random.RandomState(int(seed))
    if mode == "rw":
        return random_walk(n, rng)
    if mode == "confined_rw":
        return confined_random_walk(n, rng)
    if mode == "knot":
        return trefoil_knot(n)
    if mode == "self_avoiding_rw":
        return self_avoiding_random_walk(n, rng)
    if mode == "circle":
        return circle(n)
    if mode == "helix":
        return helix(n)
    if mode == "spiral":
        return spiral(n)
    if mode == "sphere":
        return sphere(n, rng)
    if mode == "hilbert":
        return hilbert_points(n).astype(np.float64)
    raise ValueError(f"Invalid option for initial structure: {mode!r}. Choose one of: rw, confined_rw, knot, "
                     f"self_avoiding_rw, circle, helix, spiral, sphere, hilbert.")
