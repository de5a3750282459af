"""config.ini reader for the minimizer path.

Accepts the same ``[Main]`` keys as the reference (``run.py:334-376`` upper-cases every key,
``config.py:94-312`` gives types/defaults) for everything the path consumes, applies the
``MODELLING_LEVEL`` presets of ``ArgumentChanger.convenient_argument_changer`` (``run.py:128-213``)
and adds ``PLATFORM = MI355X`` plus the engine-only key ``NB_CUTOFF``.  Quantity strings such as
``"0.1 nanometer"`` (``config.py:24-49``) reduce to floats in nm / kJ/mol / rad.
"""
from __future__ import annotations

import configparser
import dataclasses
import re
from dataclasses import dataclass, field
from typing import Any, Optional

from .system import ForceFieldParams, form_index

_BOOL_TRUE = {"true", "1", "yes", "on"}
_BOOL_FALSE = {"false", "0", "no", "off", "", "none"}


def parse_quantity(text: Any) -> float:
    """'300000.0 kilojoules_per_mole/nanometer**2' -> 300000.0 (leading number, OpenMM base units)."""
    if isinstance(text, (int, float)):
        return float(text)
    m = re.match(r"\s*([-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?)", str(text))
    if not m:
        raise ValueError(f"cannot parse quantity {text!r}")
    val = float(m.group(1))
    unit = str(text)[m.end():].strip().lower()
    if unit.startswith("angstrom"):
        val *= 0.1
    elif unit.startswith("femtosecond"):
        val *= 1e-3
    elif unit.startswith("nanosecond"):
        val *= 1e3
    elif unit.startswith("degree"):
        val *= 3.141592653589793 / 180.0
    return val


def parse_bool(text: Any) -> bool:
    if isinstance(text, bool):
        return text
    t = str(text).strip().lower()
    if t in _BOOL_TRUE:
        return True
    if t in _BOOL_FALSE:
        return False
    raise ValueError(f"cannot parse boolean {text!r}")


NOCUTOFF_MAX_BEADS = 50000   # exact NoCutoff up to here when the ini gives no NB_CUTOFF (789 iterations/s at 50 000 beads)


@dataclass
class SimulationConfig:
    """Subset of the reference's SimulationConfig that the minimizer path reads."""

    PLATFORM: str = "MI355X"
    DEVICE: int = 0
    # engine-only key, named after the OpenMM platform property ("DeterministicForces", false by default there too):
    # True selects the full-shell pair kernel, whose summation order is fixed (bitwise reproducible runs)
    DETERMINISTIC_FORCES: bool = False
    NB_CUTOFF_AUTO: bool = False   # set by load_config: the ini did not name NB_CUTOFF (see there)
    MODELLING_LEVEL: str = ""
    N_BEADS: int = 50000
    OUT_PATH: str = "results"
    INITIAL_STRUCTURE_TYPE: str = "hilbert"
    INITIAL_STRUCTURE_PATH: Optional[str] = None
    BUILD_INITIAL_STRUCTURE: bool = True
    LOOPS_PATH: str = ""
    COMPARTMENT_PATH: Optional[str] = None
    CHROM: Optional[str] = None
    LOC_START: Optional[int] = None
    LOC_END: Optional[int] = None
    SHUFFLING_SEED: int = 0
    SHUFFLE_CHROMS: bool = False            # config.py:185 (parsers: chromosome order shuffled with SHUFFLING_SEED)
    DOWNSAMPLING_PROB: float = 1.0          # config.py:157: probability of keeping a loop
    COMPARTMENT_FLIP_PROB: float = 0.0      # config.py:146-149
    COMPARTMENT_NOISE_STD: float = 0.0      # config.py:150-153
    GENE_TSV: Optional[str] = None          # config.py:168-171 (the reference ships a default table; none travels here)
    GENE_NAME: Optional[str] = None         # config.py:172
    GENE_ID: Optional[str] = None           # config.py:173
    GENE_WINDOW: int = 100000               # config.py:174
    N_ENSEMBLE: Optional[int] = None
    GENERATE_ENSEMBLE: bool = False    # config.py:142-145: run N_ENSEMBLE replicas instead of one (run.py:471)
    SIM_RUN_MD: bool = False
    SIM_N_STEPS: int = 10000           # config.py:253
    SIM_SAMPLING_STEP: int = 100       # config.py:257
    SIM_INTEGRATOR_TYPE: str = "langevin"   # config.py:258; MI355X provides langevin / verlet / brownian / amd
    SIM_INTEGRATOR_STEP: float = 0.001      # ps  ("1 femtosecond", config.py:259)
    SIM_FRICTION_COEFF: float = 0.5         # 1/ps (config.py:260-262)
    SIM_TEMPERATURE: float = 310.0          # K   (config.py:266)
    SIM_AMD_ALPHA: float = 100.0            # kJ/mol (config.py:255)
    SIM_AMD_E: float = 1000.0               # kJ/mol (config.py:256)
    TRJ_FRAMES: int = 2000                  # config.py:267
    MIN_TOLERANCE: float = 10.0        # OpenMM minimizeEnergy() default, kJ/mol/nm
    MIN_MAX_ITERATIONS: int = 0        # 0 = until converged
    ff: ForceFieldParams = field(default_factory=ForceFieldParams)

    def apply_modelling_level(self) -> None:
        """Presets of run.py:137-213 (only the keys on this path)."""
        level = "" if self.MODELLING_LEVEL is None else str(self.MODELLING_LEVEL).strip().lower()
        if level == "none":
            level = ""
        has_comp = bool(self.COMPARTMENT_PATH)
        ff = self.ff
        if level == "gene":
            self.N_BEADS = 1000
            ff.SC_USE_SPHERICAL_CONTAINER = ff.CHB_USE_CHROMOSOMAL_BLOCKS = False
            ff.SCB_USE_SUBCOMPARTMENT_BLOCKS = ff.COB_USE_COMPARTMENT_BLOCKS = False
            ff.IBL_USE_B_LAMINA_INTERACTION = ff.CF_USE_CENTRAL_FORCE = False
            self.SHUFFLE_CHROMS = False  # run.py:152
            self.SIM_RUN_MD = True
            self.SIM_N_STEPS = 10000     # run.py:154 (every preset overwrites the step count)
        elif level in ("region", "loc", "chromosome", "chrom"):
            self.N_BEADS = 5000 if level in ("region", "loc") else 20000
            ff.SC_USE_SPHERICAL_CONTAINER = ff.CHB_USE_CHROMOSOMAL_BLOCKS = False
            ff.SCB_USE_SUBCOMPARTMENT_BLOCKS = False
            ff.COB_USE_COMPARTMENT_BLOCKS = has_comp
            ff.IBL_USE_B_LAMINA_INTERACTION = ff.CF_USE_CENTRAL_FORCE = False
            self.SIM_RUN_MD = True
            self.SIM_N_STEPS = 10000     # run.py:172, 190
        elif level in ("gw", "genome"):
            self.N_BEADS = 200000
            ff.SC_USE_SPHERICAL_CONTAINER = True
            ff.CHB_USE_CHROMOSOMAL_BLOCKS = ff.SCB_USE_SUBCOMPARTMENT_BLOCKS = False
            ff.COB_USE_COMPARTMENT_BLOCKS = has_comp
            ff.IBL_USE_B_LAMINA_INTERACTION = has_comp
            ff.CF_USE_CENTRAL_FORCE = False
            self.SIM_RUN_MD = False
            self.SIM_N_STEPS = 10000     # run.py:211
        elif level:
            raise ValueError(f"unknown MODELLING_LEVEL {self.MODELLING_LEVEL!r}")


_FF_FIELDS = {f.name: f.type for f in dataclasses.fields(ForceFieldParams)}


def load_config(path_or_dict) -> SimulationConfig:
    """Reads ``[Main]`` of an ini file (or a flat dict) into a SimulationConfig; unknown keys that belong
    to parts of MultiMM outside this path (plots, nucleosomes, ...) are ignored."""
    if isinstance(path_or_dict, dict):
        raw = {str(k).upper(): v for k, v in path_or_dict.items()}
    else:
        cp = configparser.ConfigParser()
        cp.optionxform = str
        if not cp.read(path_or_dict):
            raise FileNotFoundError(path_or_dict)
        raw = {k.upper(): v for k, v in cp["Main"].items()}
    cfg = SimulationConfig()
    ffkw = {}
    for key, val in raw.items():
        if key in _FF_FIELDS:
            cur = getattr(cfg.ff, key)
            if isinstance(cur, str):  # *_FORCE_TYPE: validated like model.py:214-215 ("Unknown EV_FORCE_TYPE: ...")
                form_index(key, str(val).strip())
                ffkw[key] = str(val).strip()
            else:
                ffkw[key] = parse_bool(val) if isinstance(cur, bool) else parse_quantity(val)
        elif hasattr(cfg, key) and key != "ff":
            cur = getattr(cfg, key)
            sval = None if (isinstance(val, str) and val.strip().lower() in ("", "none")) else val
            if isinstance(cur, bool):
                setattr(cfg, key, parse_bool(val))
            elif key in ("N_BEADS", "SHUFFLING_SEED", "MIN_MAX_ITERATIONS", "DEVICE", "SIM_N_STEPS",
                         "SIM_SAMPLING_STEP", "TRJ_FRAMES", "GENE_WINDOW"):
                setattr(cfg, key, int(float(sval)) if sval is not None else getattr(cfg, key))
            elif key in ("LOC_START", "LOC_END", "N_ENSEMBLE"):
                setattr(cfg, key, int(float(sval)) if sval is not None else None)
            elif key in ("MIN_TOLERANCE", "SIM_INTEGRATOR_STEP", "SIM_FRICTION_COEFF", "SIM_TEMPERATURE",
                         "SIM_AMD_ALPHA", "SIM_AMD_E", "DOWNSAMPLING_PROB", "COMPARTMENT_FLIP_PROB",
                         "COMPARTMENT_NOISE_STD"):
                setattr(cfg, key, parse_quantity(val))
            else:
                setattr(cfg, key, sval if sval is None else str(sval))
    cfg.ff = dataclasses.replace(cfg.ff, **ffkw)
    cfg.apply_modelling_level()
    # NB_CUTOFF not given (every ini written for the reference): follow the reference -- OpenMM NoCutoff, every pair
    # (model.py:181-217 never sets a cutoff) -- as long as the exact all-pairs kernel is cheap; above 50 000 beads the
    # cell-list kernels with plain truncation at 0.6 nm take over, and MultiMM.add_forcefield says so in the log.
    cfg.NB_CUTOFF_AUTO = "NB_CUTOFF" not in raw
    if cfg.NB_CUTOFF_AUTO:
        cfg.ff = dataclasses.replace(cfg.ff, NB_CUTOFF=0.0 if int(cfg.N_BEADS) <= NOCUTOFF_MAX_BEADS else 0.6)
    return cfg
