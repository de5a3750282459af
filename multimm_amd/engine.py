"""ctypes binding of libmmx.so (C ABI: include/mmx.h) -- the device engine behind ``PLATFORM = MI355X``.

The engine plays the part OpenMM's Context/Platform plays for the reference
(``model.py:863-892``): it owns the system on one GPU, evaluates energies/forces and runs the
L-BFGS minimizer.  Failures surface as ``MMXError`` (the analogue of ``openmm.OpenMMException`` that
``bridge.py:65`` catches); there is no CPU fallback in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .system import FORM_SELECTORS, ChromatinSystem, form_index

N_TERMS = 9
N_KERNELS = 8
TERM_NAMES = ("ev", "gauss", "bond", "angle", "loop", "container", "lamina", "central", "chb")
KERNEL_NAMES = ("cell_build", "nonbonded", "backbone", "loops", "confine", "lbfgs", "reduce", "chb")
K_CELL_BUILD, K_NONBONDED, K_BACKBONE, K_LOOPS, K_CONFINE, K_LBFGS, K_REDUCE = range(7)
K_FORCES = 100   # time_kernel only: one whole force evaluation as the minimizer launches it
K_DD_LISTS = 101  # time_kernel only, decomposed handles: the halo's own kernels of one evaluation (no collective)
COMP_COB, COMP_SCB = 0, 1

# MMX_LIB: A/B timing of two builds of the library on the same box (scripts/); not a fallback mechanism
_LIB_PATH = os.environ.get("MMX_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmmx.so")


class MMXError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libmmx error {code}: {message}")
        self.code = code


class MMXStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("evaluations", C.c_int32), ("status", C.c_int32), ("n_beads", C.c_int32),
        ("e_initial", C.c_double), ("e_final", C.c_double), ("gnorm_final", C.c_double),
        ("xnorm_final", C.c_double), ("rms_force", C.c_double), ("seconds", C.c_double),
        ("energy_terms", C.c_double * N_TERMS),
        ("kernel_ns", C.c_double * N_KERNELS), ("kernel_samples", C.c_int64 * N_KERNELS),
        ("kernel_launches", C.c_int64 * N_KERNELS),
    ]

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k in ("iterations", "evaluations", "status", "n_beads", "e_initial",
                                           "e_final", "gnorm_final", "xnorm_final", "rms_force", "seconds")}
        d["energy_terms"] = {TERM_NAMES[i]: self.energy_terms[i] for i in range(N_TERMS)}
        d["kernel_us_mean"] = {
            KERNEL_NAMES[i]: (self.kernel_ns[i] / self.kernel_samples[i] / 1e3 if self.kernel_samples[i] else None)
            for i in range(N_KERNELS)}
        d["kernel_launches"] = {KERNEL_NAMES[i]: self.kernel_launches[i] for i in range(N_KERNELS)}
        return d


class MMXMdStats(C.Structure):
    """mmx_md_stats of include/mmx.h."""
    _fields_ = [("step_count", C.c_int64), ("n_steps", C.c_int32), ("integrator", C.c_int32),
                ("potential", C.c_double), ("kinetic", C.c_double), ("temperature", C.c_double),
                ("seconds", C.c_double), ("energy_terms", C.c_double * N_TERMS)]


INTEGRATORS = {"langevin": 0, "verlet": 1, "brownian": 2, "amd": 3}  # MMX_INT_*; model.py:768-808
BEAD_MASS_AMU = 16427.889                                   # forcefields/ff.xml:5

_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); also the list the symbol-export test checks against include/mmx.h
_P = C.c_void_p
SIGNATURES = {
    "mmx_abi_version": (C.c_int, []),
    "mmx_create": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(_P)]),
    "mmx_create_dd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "mmx_dd_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                              C.POINTER(C.c_int32)]),
    "mmx_dd_owned_beads": (C.c_int, [_P, _P]),
    "mmx_comm_unique_id": (C.c_int, [_P]),
    "mmx_comm_init": (C.c_int, [_P, _P]),
    "mmx_comm_init_local": (C.c_int, [C.POINTER(_P), C.c_int32]),
    "mmx_destroy": (C.c_int, [_P]),
    "mmx_last_error": (C.c_char_p, [_P]),
    "mmx_set_positions": (C.c_int, [_P, _P]),
    "mmx_get_positions": (C.c_int, [_P, _P]),
    "mmx_set_labels": (C.c_int, [_P, _P]),
    "mmx_set_backbone": (C.c_int, [_P, _P, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                   C.c_int32]),
    "mmx_set_backbone_masks": (C.c_int, [_P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32]),
    "mmx_set_loops": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_float]),
    "mmx_set_excluded_volume": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]),
    "mmx_set_compartments": (C.c_int, [_P, C.c_int32, _P, C.c_float, C.c_float]),
    "mmx_set_container": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, _P]),
    "mmx_set_lamina": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, _P]),
    "mmx_set_central": (C.c_int, [_P, C.c_float, C.c_float, _P, _P]),
    "mmx_set_chromosomal_blocks": (C.c_int, [_P, C.c_float, C.c_float, _P]),
    "mmx_set_functional_form": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "mmx_disable_term": (C.c_int, [_P, C.c_int32]),
    "mmx_set_option": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "mmx_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double)]),
    "mmx_compute": (C.c_int, [_P, _P, _P]),
    "mmx_minimize": (C.c_int, [_P, C.c_double, C.c_int32, C.POINTER(MMXStats)]),
    "mmx_md_configure": (C.c_int, [_P, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint64]),
    "mmx_md_set_amd": (C.c_int, [_P, C.c_double, C.c_double]),
    "mmx_md_set_velocities_to_temperature": (C.c_int, [_P, C.c_double, C.c_uint64]),
    "mmx_set_velocities": (C.c_int, [_P, _P]),
    "mmx_get_velocities": (C.c_int, [_P, _P]),
    "mmx_md_step": (C.c_int, [_P, C.c_int32, C.POINTER(MMXMdStats)]),
    "mmx_time_kernel": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mmx_nb_census": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mmx_cluster_census": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]),
}


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Loads libmmx.so (never builds, never falls back).  Raises if the HIP extension is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise MMXError(-2, f"{p} not found: build it with `python -m multimm_amd.build` "
                           "(hipcc, gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    """One minimizer context on one MI355X."""

    def __init__(self, n_beads: int, device: int = 0, rank: int = 0, world: int = 1):
        """``world > 1``: domain-decomposed run, this handle owns slice ``rank`` of the beads (mmx.h)."""
        self._lib = load_library()
        self._h = _P()
        self.n = int(n_beads)
        self.rank, self.world = int(rank), int(world)
        if world > 1:
            rc = self._lib.mmx_create_dd(self.n, self.rank, self.world, int(device), C.byref(self._h))
        else:
            rc = self._lib.mmx_create(self.n, int(device), C.byref(self._h))
        if rc != 0:
            msg = self._lib.mmx_last_error(None)
            raise MMXError(rc, msg.decode() if msg else "mmx_create failed")

    # what this handle owns NOW: a decomposed minimization re-assigns 62-bead segments to the ranks (mmx.h, "multi-GPU")
    def _dd_info(self):
        lo, no = C.c_int32(), C.c_int32()
        self._lib.mmx_dd_info(self._h, C.byref(lo), C.byref(no), None, None)
        return lo.value, no.value

    @property
    def own_lo(self) -> int:
        """First owned bead (the whole owned range while the ownership is the initial contiguous one)."""
        return self._dd_info()[0]

    @property
    def n_own(self) -> int:
        return self._dd_info()[1]

    def owned_beads(self) -> np.ndarray:
        """Global ids of the owned beads in local order (ascending): row k of ``compute()``'s forces belongs to bead
        ``owned_beads()[k]``."""
        ids = np.empty(self.n_own, dtype=np.int32)
        self._chk(self._lib.mmx_dd_owned_beads(self._h, ids.ctypes.data_as(_P)))
        return ids

    @staticmethod
    def comm_unique_id() -> bytes:
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        rc = lib.mmx_comm_unique_id(C.cast(buf, _P))
        if rc != 0:
            raise MMXError(rc, (lib.mmx_last_error(None) or b"?").decode())
        return bytes(buf)

    @staticmethod
    def comm_init_local(engines):
        """Loopback communicator over ranks 0..world-1 living in this process on one device (mmx.h); afterwards
        drive every engine from its own thread."""
        lib = load_library()
        arr = (_P * len(engines))(*[e._h for e in engines])
        rc = lib.mmx_comm_init_local(arr, len(engines))
        if rc != 0:
            raise MMXError(rc, (lib.mmx_last_error(engines[0]._h) or b"?").decode())

    def comm_init(self, unique_id: bytes):
        """Collective over all ranks: RCCL communicator on this handle's device."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._chk(self._lib.mmx_comm_init(self._h, C.cast(buf, _P)))

    # -- plumbing ------------------------------------------------------------------------------
    def _chk(self, rc: int):
        if rc != 0:
            msg = self._lib.mmx_last_error(self._h)
            raise MMXError(rc, msg.decode() if msg else "?")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mmx_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- system description --------------------------------------------------------------------
    def set_positions(self, xyz_nm):
        a = _f32(xyz_nm).reshape(self.n, 3)
        self._chk(self._lib.mmx_set_positions(self._h, a.ctypes.data))

    def get_positions(self) -> np.ndarray:
        out = np.empty((self.n, 3), dtype=np.float32)
        self._chk(self._lib.mmx_get_positions(self._h, out.ctypes.data))
        return out

    def set_labels(self, labels):
        a = np.ascontiguousarray(labels, dtype=np.int8)
        if a.shape != (self.n,):
            raise ValueError("labels must be [N]")
        self._chk(self._lib.mmx_set_labels(self._h, a.ctypes.data))

    def set_backbone(self, chr_ends, r0, k_bond, theta0, k_angle, use_bond=True, use_angle=True):
        a = np.ascontiguousarray(chr_ends, dtype=np.int32)
        self._chk(self._lib.mmx_set_backbone(self._h, a.ctypes.data, len(a), r0, k_bond, theta0, k_angle,
                                             int(use_bond), int(use_angle)))

    def set_backbone_masks(self, flags, r0, k_bond, theta0, k_angle, use_bond=True, use_angle=True):
        a = np.ascontiguousarray(flags, dtype=np.uint8)
        if a.shape != (self.n,):
            raise ValueError("flags must be [N]")
        self._chk(self._lib.mmx_set_backbone_masks(self._h, a.ctypes.data, r0, k_bond, theta0, k_angle,
                                                   int(use_bond), int(use_angle)))

    def set_loops(self, m, n, r0, k_loop):
        m = np.ascontiguousarray(m, dtype=np.int32)
        n = np.ascontiguousarray(n, dtype=np.int32)
        r0 = _f32(r0)
        if not (len(m) == len(n) == len(r0)):
            raise ValueError("loop arrays must have equal length")
        self._chk(self._lib.mmx_set_loops(self._h, m.ctypes.data, n.ctypes.data, r0.ctypes.data, len(m), k_loop))

    def set_excluded_volume(self, eps, sigma, r_small, power, cutoff_nm):
        self._chk(self._lib.mmx_set_excluded_volume(self._h, eps, sigma, r_small, power, cutoff_nm))

    def set_compartments(self, mode, E, rc, cutoff_nm):
        e = _f32(E)
        if len(e) != (2 if mode == COMP_COB else 4):
            raise ValueError("COB takes E[2], SCB takes E[4]")
        self._chk(self._lib.mmx_set_compartments(self._h, mode, e.ctypes.data, rc, cutoff_nm))

    def set_container(self, C_, R1, R2, centre):
        c = _f32(centre)
        self._chk(self._lib.mmx_set_container(self._h, C_, R1, R2, c.ctypes.data))

    def set_lamina(self, B, R1, R2, centre):
        c = _f32(centre)
        self._chk(self._lib.mmx_set_lamina(self._h, B, R1, R2, c.ctypes.data))

    def set_central(self, G, R1, centre, w):
        c = _f32(centre)
        ww = _f32(w)
        if ww.shape != (self.n,):
            raise ValueError("w must be [N]")
        self._chk(self._lib.mmx_set_central(self._h, G, R1, c.ctypes.data, ww.ctypes.data))

    def set_chromosomal_blocks(self, k_C, dE, chrom):
        c = np.ascontiguousarray(chrom, dtype=np.int32)
        if c.shape != (self.n,):
            raise ValueError("chrom must be [N]")
        self._chk(self._lib.mmx_set_chromosomal_blocks(self._h, k_C, dE, c.ctypes.data))

    def set_functional_form(self, selector: int, form: int):
        self._chk(self._lib.mmx_set_functional_form(self._h, int(selector), int(form)))

    def disable_term(self, term: int):
        self._chk(self._lib.mmx_disable_term(self._h, term))

    def set_option(self, key: str, value: float):
        self._chk(self._lib.mmx_set_option(self._h, key.encode(), float(value)))

    def get_option(self, key: str) -> float:
        v = C.c_double()
        self._chk(self._lib.mmx_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    # -- compute / minimize ----------------------------------------------------------------------
    def compute(self, forces: bool = True):
        """Returns (energy_terms[8] float64, forces [N,3] float32 or None) at the current positions."""
        et = np.zeros(N_TERMS, dtype=np.float64)
        f = np.empty((self.n_own, 3), dtype=np.float32) if forces else None
        self._chk(self._lib.mmx_compute(self._h, f.ctypes.data if forces else None, et.ctypes.data))
        return et, f

    def minimize(self, tolerance: float = 10.0, max_iters: int = 0) -> MMXStats:
        st = MMXStats()
        self._chk(self._lib.mmx_minimize(self._h, float(tolerance), int(max_iters), C.byref(st)))
        return st

    # -- molecular dynamics (model.py:768-808, 878, 907-995) -------------------------------------------
    def md_configure(self, integrator: str = "langevin", dt_ps: float = 0.001, temperature_K: float = 310.0,
                     friction_per_ps: float = 0.5, mass_amu: float = BEAD_MASS_AMU, seed: int = 0,
                     amd_alpha: float = 100.0, amd_e: float = 1000.0):
        """``integrator`` "amd" = mm.amd.AMDIntegrator(dt, amd_alpha, amd_e) (model.py:794-800; kJ/mol, defaults of
        config.py:255-256); temperature and friction are ignored by it and by "verlet"."""
        if integrator not in INTEGRATORS:
            raise MMXError(-1, f"integrator {integrator!r} is not provided by the MI355X engine "
                               f"(available: {', '.join(INTEGRATORS)})")
        self._chk(self._lib.mmx_md_configure(self._h, INTEGRATORS[integrator], float(dt_ps), float(temperature_K),
                                             float(friction_per_ps), float(mass_amu), int(seed)))
        if integrator == "amd":
            self._chk(self._lib.mmx_md_set_amd(self._h, float(amd_alpha), float(amd_e)))

    def set_velocities_to_temperature(self, temperature_K: float, seed: int = 0):
        self._chk(self._lib.mmx_md_set_velocities_to_temperature(self._h, float(temperature_K), int(seed)))

    def set_velocities(self, v_nm_per_ps):
        a = _f32(v_nm_per_ps).reshape(self.n, 3)
        self._chk(self._lib.mmx_set_velocities(self._h, a.ctypes.data))

    def get_velocities(self) -> np.ndarray:
        out = np.zeros((self.n, 3), dtype=np.float32)
        self._chk(self._lib.mmx_get_velocities(self._h, out.ctypes.data))
        return out

    def md_step(self, n_steps: int) -> MMXMdStats:
        st = MMXMdStats()
        self._chk(self._lib.mmx_md_step(self._h, int(n_steps), C.byref(st)))
        return st

    def time_kernel(self, kernel: int, reps: int = 20):
        us, by = C.c_double(), C.c_double()
        self._chk(self._lib.mmx_time_kernel(self._h, kernel, reps, C.byref(us), C.byref(by)))
        return us.value, by.value

    def nb_census(self) -> dict:
        nc, mx = C.c_int64(), C.c_int32()
        edge, cand, within = C.c_double(), C.c_double(), C.c_double()
        self._chk(self._lib.mmx_nb_census(self._h, C.byref(nc), C.byref(mx), C.byref(edge), C.byref(cand),
                                          C.byref(within)))
        return dict(n_cells=nc.value, max_per_cell=mx.value, cell_edge=edge.value, pair_candidates=cand.value,
                    pairs_within_cutoff=within.value)

    def cluster_census(self) -> dict:
        nc = C.c_int64()
        cand, acc, beads = C.c_double(), C.c_double(), C.c_double()
        self._chk(self._lib.mmx_cluster_census(self._h, C.byref(nc), C.byref(cand), C.byref(acc), C.byref(beads)))
        return dict(n_clusters=nc.value, tiles_candidate=cand.value, tiles_accepted=acc.value,
                    beads_swept=beads.value)

    # -- convenience: upload a whole ChromatinSystem the way add_forcefield orders it ---------------
    def load_system(self, s: ChromatinSystem):
        """Installs every enabled term of ``s`` in the order of ``add_forcefield`` (model.py:812-857)."""
        if s.n_beads != self.n:
            raise ValueError("system size does not match the engine")
        ff = s.ff
        R1, R2, r_comp = s.radii
        centre = s.centre
        self.set_positions(s.positions)
        self.set_labels(s.labels)
        if ff.EV_USE_EXCLUDED_VOLUME:  # sigma is LE_HARMONIC_BOND_R0 (sic), model.py:175
            self.set_excluded_volume(ff.EV_EPSILON, ff.LE_HARMONIC_BOND_R0, ff.EV_R_SMALL, ff.EV_POWER, ff.NB_CUTOFF)
        if ff.COB_USE_COMPARTMENT_BLOCKS:
            self.set_compartments(COMP_COB, [ff.COB_EA, ff.COB_EB], r_comp, ff.NB_CUTOFF)
        if ff.SCB_USE_SUBCOMPARTMENT_BLOCKS:
            self.set_compartments(COMP_SCB, [ff.SCB_EA1, ff.SCB_EA2, ff.SCB_EB1, ff.SCB_EB2], r_comp, ff.NB_CUTOFF)
        if ff.CHB_USE_CHROMOSOMAL_BLOCKS:
            self.set_chromosomal_blocks(ff.CHB_KC, ff.CHB_DE, s.chrom_of)
        if ff.SC_USE_SPHERICAL_CONTAINER:
            self.set_container(ff.SC_SCALE, R1, R2, centre)
        if ff.IBL_USE_B_LAMINA_INTERACTION:
            self.set_lamina(ff.IBL_SCALE, R1, R2, centre)
        if ff.CF_USE_CENTRAL_FORCE:
            if s.chrom_strength is None:
                raise ValueError("central force needs chrom_strength")
            self.set_central(ff.CF_STRENGTH, R1, centre, s.chrom_strength)
        if ff.POL_USE_HARMONIC_BOND or ff.POL_USE_HARMONIC_ANGLE:
            self.set_backbone(s.chr_ends, ff.POL_HARMONIC_BOND_R0, ff.POL_HARMONIC_BOND_K,
                              ff.POL_HARMONIC_ANGLE_R0, ff.POL_HARMONIC_ANGLE_CONSTANT_K,
                              ff.POL_USE_HARMONIC_BOND, ff.POL_USE_HARMONIC_ANGLE)
        if ff.LE_USE_HARMONIC_BOND and s.n_loops:
            self.set_loops(s.loop_m, s.loop_n, s.loop_rest_lengths(), ff.LE_HARMONIC_BOND_K)
        for key, sel in FORM_SELECTORS.items():  # the *_FORCE_TYPE keys (config.py:269-312)
            self.set_functional_form(sel, form_index(key, getattr(ff, key)))
        return self


def engine_for(system: ChromatinSystem, device: int = 0, rank: int = 0, world: int = 1) -> Engine:
    return Engine(system.n_beads, device, rank, world).load_system(system)
