"""System description for the minimizer path: the tensors and constants MultiMM hands to OpenMM.

Mirrors what ``MultiMM.__init__`` / ``add_forcefield`` assemble in the reference
(``src/multimm/model.py:25-162, 812-857``): bead count, start positions, compartment labels ``Cs``,
chromosome boundaries ``chr_ends``, loop anchors ``ms, ns`` with rest lengths ``ds``, and the
force-field constants of ``config.py:188-247``.  Data-file parsers are out of scope (SURVEY.md
section 8 f2); ``synthetic_system`` generates inputs with the same tensor contracts.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Optional

import numpy as np

from .hilbert import hilbert_points

# hg38 chromosome lengths used by the reference to place chromosome boundaries (utils.py:67-95).
CHROM_LENGTHS = np.array(
    [248387328, 242696752, 201105948, 193574945, 182045439, 172126628, 160567428, 146259331, 150617247,
     134758134, 135127769, 133324548, 113566686, 101161492, 99753195, 96330374, 84276897, 80542538,
     61707364, 66210255, 45090682, 51324926, 154259566, 62460029], dtype=np.int64)


@dataclass
class ForceFieldParams:
    """Force-field switches and constants; names and defaults follow ``config.py:188-247, 269-312``."""

    POL_USE_HARMONIC_BOND: bool = True
    POL_HARMONIC_BOND_R0: float = 0.1          # nm
    POL_HARMONIC_BOND_K: float = 300000.0      # kJ/mol/nm^2
    POL_USE_HARMONIC_ANGLE: bool = True
    POL_HARMONIC_ANGLE_R0: float = float(np.pi)  # rad
    POL_HARMONIC_ANGLE_CONSTANT_K: float = 100.0
    LE_USE_HARMONIC_BOND: bool = True
    LE_FIXED_DISTANCES: bool = False
    LE_HARMONIC_BOND_R0: float = 0.1
    LE_HARMONIC_BOND_K: float = 30000.0
    EV_USE_EXCLUDED_VOLUME: bool = True
    EV_EPSILON: float = 100.0
    EV_R_SMALL: float = 0.05
    EV_POWER: float = 6.0
    SC_USE_SPHERICAL_CONTAINER: bool = False
    SC_SCALE: float = 1000.0
    COB_USE_COMPARTMENT_BLOCKS: bool = False
    COB_EA: float = 1.0
    COB_EB: float = 2.0
    SCB_USE_SUBCOMPARTMENT_BLOCKS: bool = False
    SCB_EA1: float = 1.0
    SCB_EA2: float = 1.33
    SCB_EB1: float = 1.66
    SCB_EB2: float = 2.0
    IBL_USE_B_LAMINA_INTERACTION: bool = False
    IBL_SCALE: float = 400.0
    CF_USE_CENTRAL_FORCE: bool = False
    CF_STRENGTH: float = 20.0
    CHB_USE_CHROMOSOMAL_BLOCKS: bool = False
    CHB_KC: float = 0.3
    CHB_DE: float = 1e-4
    # Functional forms (config.py:269-312); the first entry of FORM_NAMES[key] is the default
    EV_FORCE_TYPE: str = "powerlaw"
    COB_FORCE_TYPE: str = "gaussian"
    SCB_FORCE_TYPE: str = "gaussian"
    CHB_FORCE_TYPE: str = "polynomial"
    BLAMINA_FORCE_TYPE: str = "sin"
    CENTRAL_FORCE_TYPE: str = "harmonic"
    LE_LOOP_FORCE_TYPE: str = "harmonic"
    # Engine-only key: pair cutoff in nm.  <= 0 reproduces the reference (OpenMM NoCutoff, all pairs);
    # > 0 is plain truncation (OpenMM CutoffNonPeriodic) and selects the cell-list kernel.
    NB_CUTOFF: float = 0.6


# ini key -> (names in the order of mmx_set_functional_form's form index, MMX_SEL_* selector)
FORM_NAMES = {
    "EV_FORCE_TYPE": ("powerlaw", "gaussian_core"),
    "COB_FORCE_TYPE": ("gaussian", "yukawa", "theta"),
    "SCB_FORCE_TYPE": ("gaussian", "yukawa", "theta"),
    "CHB_FORCE_TYPE": ("polynomial", "gaussian", "saturating"),
    "BLAMINA_FORCE_TYPE": ("sin", "gaussian_shell", "harmonic_shell", "logistic_shell"),
    "CENTRAL_FORCE_TYPE": ("harmonic", "gaussian", "logistic"),
    "LE_LOOP_FORCE_TYPE": ("harmonic", "fene_soft", "gaussian_tether"),
}
FORM_SELECTORS = {"EV_FORCE_TYPE": 0, "COB_FORCE_TYPE": 1, "SCB_FORCE_TYPE": 2, "CHB_FORCE_TYPE": 3,
                  "BLAMINA_FORCE_TYPE": 4, "CENTRAL_FORCE_TYPE": 5, "LE_LOOP_FORCE_TYPE": 6}


def form_index(key: str, name: str) -> int:
    """Index of a *_FORCE_TYPE value; unknown names raise like the reference does (model.py:214-215 etc.)."""
    names = FORM_NAMES[key]
    if str(name) not in names:
        raise ValueError(f"Unknown {key}: {name}")
    return names.index(str(name))


def set_radiuses(n_beads: int, b0: float) -> tuple[float, float, float]:
    """``MultiMM.set_radiuses`` (model.py:1016-1067): returns (R1, R2, r_comp) in nm."""
    R2 = b0 * float(n_beads) ** (1.0 / 3.0)
    R1 = R2 * 0.20 ** (1.0 / 3.0)
    return R1, R2, 1.5 * b0


def backbone_flags(n_beads: int, chr_ends: np.ndarray) -> np.ndarray:
    """Per-bead masks reproducing the reference's index quirks.

    bit0: bond (i,i+1) exists  <=> i in [0,N-2] and i not in chr_ends                 (model.py:628-629)
    bit1: angle (i,i+1,i+2) exists <=> i in [0,N-3], i not in chr_ends, i not in chr_ends-1 (model.py:711-712)
    """
    ce = np.asarray(chr_ends, dtype=np.int64)
    i = np.arange(n_beads)
    in_ends = np.isin(i, ce)
    in_ends_m1 = np.isin(i, ce - 1)
    f = np.zeros(n_beads, dtype=np.uint8)
    f[(i <= n_beads - 2) & ~in_ends] |= 1
    f[(i <= n_beads - 3) & ~in_ends & ~in_ends_m1] |= 2
    return f


def gw_chr_ends(n_beads: int, n_chroms: int = 22) -> np.ndarray:
    """Genome-wide chromosome boundaries in bead space, as ``import_bed`` computes them
    (utils.py:254-273): cumsum of the first ``n_chroms`` hg38 lengths // resolution, last forced to N."""
    ends = np.cumsum(np.insert(CHROM_LENGTHS[:n_chroms], 0, 0))
    resolution = ends[-1] // n_beads
    ends = ends // resolution
    ends[-1] = n_beads
    return ends.astype(np.int32)


def chrom_strength_per_bead(chr_ends: np.ndarray, n_beads: int) -> np.ndarray:
    """``chrom_strength`` weights of the central force (utils.py:137, model.py:158-162)."""
    lens = CHROM_LENGTHS.astype(np.float64)
    cs = 1.0 - (lens - lens.min()) / (lens.max() - lens.min())
    w = np.zeros(n_beads, dtype=np.float64)
    for k in range(len(chr_ends) - 1):
        w[chr_ends[k]:chr_ends[k + 1]] = cs[k % len(cs)]
    return w


@dataclass
class ChromatinSystem:
    n_beads: int
    positions: np.ndarray                      # [N,3] float64 nm
    chr_ends: np.ndarray                       # int32, first-bead indices [0, e1, ..., N]
    labels: np.ndarray                         # [N] int8 in {-2..2} (Cs)
    loop_m: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    loop_n: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    loop_r0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))  # ds
    ff: ForceFieldParams = field(default_factory=ForceFieldParams)
    chrom_strength: Optional[np.ndarray] = None
    name: str = "custom"
    seed: int = 0

    def __post_init__(self):
        self.positions = np.ascontiguousarray(self.positions, dtype=np.float64).reshape(self.n_beads, 3)
        self.chr_ends = np.ascontiguousarray(self.chr_ends, dtype=np.int32)
        self.labels = np.ascontiguousarray(self.labels, dtype=np.int8)
        self.loop_m = np.ascontiguousarray(self.loop_m, dtype=np.int32)
        self.loop_n = np.ascontiguousarray(self.loop_n, dtype=np.int32)
        self.loop_r0 = np.ascontiguousarray(self.loop_r0, dtype=np.float64)
        if self.labels.shape != (self.n_beads,):
            raise ValueError("labels must be [N]")
        if np.any(np.abs(self.labels) > 2):
            raise ValueError("labels must lie in {-2..2}")
        if not (len(self.loop_m) == len(self.loop_n) == len(self.loop_r0)):
            raise ValueError("loop arrays must have equal length")

    # quantities the reference derives in run(): set_radiuses (model.py:1016) and mass_center (model.py:759)
    @property
    def radii(self) -> tuple[float, float, float]:
        return set_radiuses(self.n_beads, self.ff.POL_HARMONIC_BOND_R0)

    @property
    def centre(self) -> np.ndarray:
        return self.positions.mean(axis=0)

    @property
    def flags(self) -> np.ndarray:
        return backbone_flags(self.n_beads, self.chr_ends)

    @property
    def chrom_of(self) -> np.ndarray:
        """``chrom_spin`` (model.py:158-162): index of the chr_ends interval a bead lies in; only equality
        of two beads' values matters to the chromosomal-block force."""
        return (np.searchsorted(self.chr_ends, np.arange(self.n_beads), side="right") - 1).astype(np.int32)

    @property
    def n_loops(self) -> int:
        return int(len(self.loop_m))

    def loop_rest_lengths(self) -> np.ndarray:
        """r0 per loop as add_loops chooses it (model.py:657)."""
        if self.ff.LE_FIXED_DISTANCES:
            return np.full(self.n_loops, self.ff.LE_HARMONIC_BOND_R0, dtype=np.float64)
        return self.loop_r0

    def gauss_table(self) -> np.ndarray:
        """5x5 amplitude table E(s_i+2, s_j+2) of the compartment Gaussians (model.py:246-253, 322-333)."""
        t = np.zeros((5, 5), dtype=np.float64)
        ff = self.ff
        if ff.COB_USE_COMPARTMENT_BLOCKS:
            for a in (1, 2):
                for b in (1, 2):
                    t[a + 2, b + 2] += ff.COB_EA
            for a in (-1, -2):
                for b in (-1, -2):
                    t[a + 2, b + 2] += ff.COB_EB
        if ff.SCB_USE_SUBCOMPARTMENT_BLOCKS:
            t[4, 4] += ff.SCB_EA1
            t[3, 3] += ff.SCB_EA2
            t[1, 1] += ff.SCB_EB1
            t[0, 0] += ff.SCB_EB2
        return t

    def with_ff(self, **kw) -> "ChromatinSystem":
        return replace(self, ff=replace(self.ff, **kw))


# --------------------------------------------------------------------------------------------------
# Synthetic inputs: same tensor contracts as the reference's data parsers.
# --------------------------------------------------------------------------------------------------
def _synthetic_labels(n: int, rng: np.random.Generator) -> np.ndarray:
    """Piecewise-constant compartment domains, lengths ~ Geometric(mean 40), values uniform on
    {-2,-1,1,2} with 5 % zeros."""
    labels = np.zeros(n, dtype=np.int8)
    pos = 0
    while pos < n:
        length = int(rng.geometric(1.0 / 40.0))
        val = 0 if rng.random() < 0.05 else int(rng.choice([-2, -1, 1, 2]))
        labels[pos:pos + length] = val
        pos += length
    return labels


def _synthetic_loops(n: int, chr_ends: np.ndarray, n_loops: int, rng: np.random.Generator):
    """Loop anchors with the statistics of the reference's fixture: m uniform
    inside a chromosome, n = m + max(3, round(Exp(23))) kept inside that chromosome, unique pairs
    (utils.py:507), n > m + 2 (utils.py:515-519), r0 = 0.1 + 0.1*U(0,1) in [0.1, 0.2] nm (utils.py:520)."""
    pairs = set()
    ms, ns = [], []
    guard = 0
    while len(ms) < n_loops and guard < 50 * n_loops + 1000:
        guard += 1
        m = int(rng.integers(0, n - 4))
        sep = max(3, int(round(rng.exponential(23.0))))
        k = int(np.searchsorted(chr_ends, m, side="right")) - 1
        hi = int(chr_ends[min(k + 1, len(chr_ends) - 1)]) - 1
        q = m + sep
        if q > hi or q >= n or (m, q) in pairs:
            continue
        pairs.add((m, q))
        ms.append(m)
        ns.append(q)
    order = np.lexsort((ns, ms))
    ms = np.asarray(ms, dtype=np.int32)[order]
    ns = np.asarray(ns, dtype=np.int32)[order]
    r0 = 0.1 + 0.1 * rng.random(len(ms))
    return ms, ns, r0


_PRESETS = {
    # name: (N, genome-wide?, loops, force switches)
    "region_500": (500, False, 10, {}),
    "region_5k": (5000, False, 100, {}),            # BASELINE config 1 (REGION preset, run.py:164)
    "chr1_50k": (50000, False, 1000, {}),           # BASELINE config 2
    "gw_200k": (200000, True, 4235, dict(            # BASELINE config 3 (GW preset, run.py:202-213)
        SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, IBL_USE_B_LAMINA_INTERACTION=True)),
    "gw_1m": (1000000, True, 21000, dict(            # BASELINE config 5
        SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, IBL_USE_B_LAMINA_INTERACTION=True)),
}


def synthetic_system(name: str = "gw_200k", seed: int = 0, n_beads: Optional[int] = None,
                     jitter: float = 0.0, start: str = "hilbert", **ff_overrides) -> ChromatinSystem:
    """Synthetic Hilbert-curve-initialised bead system for one of BASELINE.json's configurations.

    ``n_beads`` rescales a preset (loops scale with N); ``jitter`` adds N(0, jitter nm) noise with
    seed 1234 to break lattice degeneracy; ``start`` is any INITIAL_STRUCTURE_TYPE (``multimm_amd/initial_structure.py``;
    'circle' = ``polymer_circle(N, 50, 5)`` as config_specific_region.ini uses).
    """
    if name not in _PRESETS:
        raise ValueError(f"unknown preset {name!r}; choose from {sorted(_PRESETS)}")
    n0, gw, l0, switches = _PRESETS[name]
    n = int(n_beads) if n_beads else n0
    n_loops = max(1, int(round(l0 * n / n0)))
    rng = np.random.default_rng(seed)
    from .initial_structure import compute_init_struct
    pos = compute_init_struct(n, start, seed=seed) * 0.1   # the mmCIF's Angstrom read back as nm
    if jitter > 0.0:
        pos = pos + np.random.default_rng(1234).normal(0.0, jitter, size=pos.shape)
    chr_ends = gw_chr_ends(n) if gw else np.array([0, n], dtype=np.int32)
    labels = _synthetic_labels(n, rng)
    ms, ns, r0 = _synthetic_loops(n, chr_ends, n_loops, rng)
    ff = replace(ForceFieldParams(), **switches)
    ff = replace(ff, **ff_overrides)
    sys_ = ChromatinSystem(n_beads=n, positions=pos, chr_ends=chr_ends, labels=labels, loop_m=ms, loop_n=ns,
                           loop_r0=r0, ff=ff, name=name if not n_beads else f"{name}@{n}", seed=seed)
    if ff.CF_USE_CENTRAL_FORCE:
        sys_.chrom_strength = chrom_strength_per_bead(chr_ends, n)
    return sys_
