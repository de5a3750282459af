"""ctypes wrapper of the fp64 C oracle (oracle/mmx_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from multimm_amd/.  PARITY UNPINNED (no reference golden vectors exist and
OpenMM is not installed): pinned by known-answer tests and finite differences instead.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmmx_oracle.so")
N_TERMS = 9
TERM_NAMES = ("ev", "gauss", "bond", "angle", "loop", "container", "lamina", "central", "chb")


class OrcSystem(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("labels", C.c_void_p), ("bb_flags", C.c_void_p),
        ("use_bond", C.c_int32), ("use_angle", C.c_int32),
        ("bond_r0", C.c_double), ("bond_k", C.c_double), ("angle_theta0", C.c_double), ("angle_k", C.c_double),
        ("n_loops", C.c_int32), ("loop_m", C.c_void_p), ("loop_n", C.c_void_p), ("loop_r0", C.c_void_p),
        ("loop_k", C.c_double),
        ("use_ev", C.c_int32), ("ev_eps", C.c_double), ("ev_sigma", C.c_double), ("ev_rsmall", C.c_double),
        ("ev_power", C.c_double), ("ev_cutoff", C.c_double),
        ("use_gauss", C.c_int32), ("gauss_table", C.c_double * 25), ("gauss_rc", C.c_double),
        ("gauss_cutoff", C.c_double),
        ("use_container", C.c_int32), ("sc_C", C.c_double), ("sc_R1", C.c_double), ("sc_R2", C.c_double),
        ("use_lamina", C.c_int32), ("ibl_B", C.c_double), ("ibl_R1", C.c_double), ("ibl_R2", C.c_double),
        ("use_central", C.c_int32), ("cf_G", C.c_double), ("cf_R1", C.c_double), ("cf_w", C.c_void_p),
        ("centre", C.c_double * 3),
        ("use_chb", C.c_int32), ("chb_kc", C.c_double), ("chb_de", C.c_double), ("chrom_of", C.c_void_p),
        ("ev_form", C.c_int32), ("has_cob", C.c_int32), ("has_scb", C.c_int32), ("cob_form", C.c_int32),
        ("scb_form", C.c_int32), ("tab_cob", C.c_double * 25), ("tab_scb", C.c_double * 25),
        ("chb_form", C.c_int32), ("lam_form", C.c_int32), ("cf_form", C.c_int32), ("loop_form", C.c_int32),
    ]


class OrcMinStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("evaluations", C.c_int32), ("status", C.c_int32),
                ("e_initial", C.c_double), ("e_final", C.c_double), ("gnorm_final", C.c_double),
                ("xnorm_final", C.c_double), ("seconds", C.c_double)]


class OrcMdStats(C.Structure):
    _fields_ = [("step_count", C.c_int64), ("potential", C.c_double), ("kinetic", C.c_double),
                ("temperature", C.c_double), ("eterms", C.c_double * N_TERMS)]


MD_KINDS = {"langevin": 0, "verlet": 1, "brownian": 2, "amd": 3}

# Functional forms by ini key, default first (config.py:269-312; branches of model.py:173-215, 229-292, 305-382,
# 395-449, 479-544, 557-615, 648-703).  The index is what orc_system.*_form holds.
FORM_NAMES = {
    "EV_FORCE_TYPE": ("powerlaw", "gaussian_core"),
    "COB_FORCE_TYPE": ("gaussian", "yukawa", "theta"),
    "SCB_FORCE_TYPE": ("gaussian", "yukawa", "theta"),
    "CHB_FORCE_TYPE": ("polynomial", "gaussian", "saturating"),
    "BLAMINA_FORCE_TYPE": ("sin", "gaussian_shell", "harmonic_shell", "logistic_shell"),
    "CENTRAL_FORCE_TYPE": ("harmonic", "gaussian", "logistic"),
    "LE_LOOP_FORCE_TYPE": ("harmonic", "fene_soft", "gaussian_tether"),
}


def build(force: bool = False) -> str:
    """(Re)builds the library when a source is newer.  MMX_ORACLE_LIB names another build of it to load instead (the
    sanitizer build: `make -C oracle asan`, see the Makefile) -- never a different implementation."""
    if os.environ.get("MMX_ORACLE_LIB"):
        return os.path.abspath(os.environ["MMX_ORACLE_LIB"])
    srcs = [os.path.join(HERE, f) for f in ("mmx_oracle.c", "mmx_cpu_fast.c", "mmx_cpu_fast_sweep.c", "mmx_oracle.h", "Makefile")]

    def stale():
        return force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in srcs)
    if stale():
        import fcntl
        # several processes may arrive here together (rank processes, pytest workers): one builds, the others wait and find
        # the library current (the Makefile itself builds in a private directory and renames the result into place)
        with open(os.path.join(HERE, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if stale():
                    subprocess.run(["make", "-C", HERE, "-s", "-B" if force else "-s"], check=True)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_eval.restype = C.c_int
        _lib.orc_eval.argtypes = [C.POINTER(OrcSystem), C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_minimize.restype = C.c_int
        _lib.orc_minimize.argtypes = [C.POINTER(OrcSystem), C.c_void_p, C.c_double, C.c_int, C.POINTER(OrcMinStats)]
        _lib.orc_hilbert_points.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_backbone_flags.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        _lib.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.orc_normal3.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
        _lib.orc_md_velocities.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_void_p]
        _lib.orc_md_step.restype = C.c_int
        _lib.orc_md_step.argtypes = [C.POINTER(OrcSystem), C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                     C.c_uint64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(OrcMdStats)]
        _lib.orc_md_step_amd.restype = C.c_int
        _lib.orc_md_step_amd.argtypes = [C.POINTER(OrcSystem), C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(OrcMdStats)]
        _lib.orc_fast_eval.restype = C.c_int
        _lib.orc_fast_eval.argtypes = [C.POINTER(OrcSystem), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        _lib.orc_fast_minimize.restype = C.c_int
        _lib.orc_fast_minimize.argtypes = [C.POINTER(OrcSystem), C.c_void_p, C.c_double, C.c_int, C.POINTER(OrcMinStats),
                                           C.POINTER(C.c_double)]
        assert _lib.orc_sizeof_system() == C.sizeof(OrcSystem), "OrcSystem layout mismatch"
    return _lib


class Oracle:
    """fp64 CPU evaluation of a multimm_amd.system.ChromatinSystem (duck-typed: only reads attributes)."""

    def __init__(self, system, cutoff=None, as_float32_inputs: bool = True):
        """``as_float32_inputs``: round positions / rest lengths / constants through fp32 first, so the
        oracle sees exactly the numbers the fp32 device path receives (arithmetic stays fp64)."""
        self.s = system
        ff = system.ff
        n = system.n_beads
        r32 = (lambda v: float(np.float32(v))) if as_float32_inputs else float
        R1, R2, r_comp = system.radii
        self._keep = {}
        o = OrcSystem()
        o.n = n
        self._keep["labels"] = np.ascontiguousarray(system.labels, dtype=np.int8)
        o.labels = self._keep["labels"].ctypes.data
        self._keep["flags"] = np.ascontiguousarray(system.flags, dtype=np.uint8)
        o.bb_flags = self._keep["flags"].ctypes.data
        o.use_bond = int(ff.POL_USE_HARMONIC_BOND)
        o.use_angle = int(ff.POL_USE_HARMONIC_ANGLE)
        o.bond_r0, o.bond_k = r32(ff.POL_HARMONIC_BOND_R0), r32(ff.POL_HARMONIC_BOND_K)
        o.angle_theta0, o.angle_k = r32(ff.POL_HARMONIC_ANGLE_R0), r32(ff.POL_HARMONIC_ANGLE_CONSTANT_K)
        if ff.LE_USE_HARMONIC_BOND and system.n_loops:
            self._keep["m"] = np.ascontiguousarray(system.loop_m, dtype=np.int32)
            self._keep["n"] = np.ascontiguousarray(system.loop_n, dtype=np.int32)
            r0 = np.asarray(system.loop_rest_lengths(), dtype=np.float64)
            if as_float32_inputs:
                r0 = r0.astype(np.float32).astype(np.float64)
            self._keep["r0"] = np.ascontiguousarray(r0)
            o.n_loops = system.n_loops
            o.loop_m, o.loop_n, o.loop_r0 = (self._keep[k].ctypes.data for k in ("m", "n", "r0"))
            o.loop_k = r32(ff.LE_HARMONIC_BOND_K)
        rc = ff.NB_CUTOFF if cutoff is None else cutoff
        o.use_ev = int(ff.EV_USE_EXCLUDED_VOLUME)
        o.ev_eps, o.ev_sigma = r32(ff.EV_EPSILON), r32(ff.LE_HARMONIC_BOND_R0)  # sigma quirk: model.py:175
        o.ev_rsmall, o.ev_power, o.ev_cutoff = r32(ff.EV_R_SMALL), r32(ff.EV_POWER), r32(rc) if rc > 0 else rc
        tab = system.gauss_table()
        o.use_gauss = int(ff.COB_USE_COMPARTMENT_BLOCKS or ff.SCB_USE_SUBCOMPARTMENT_BLOCKS)
        for i in range(25):
            o.gauss_table[i] = r32(tab.flat[i])
        o.gauss_rc, o.gauss_cutoff = r32(r_comp), r32(rc) if rc > 0 else rc
        o.use_container = int(ff.SC_USE_SPHERICAL_CONTAINER)
        o.sc_C, o.sc_R1, o.sc_R2 = r32(ff.SC_SCALE), r32(R1), r32(R2)
        o.use_lamina = int(ff.IBL_USE_B_LAMINA_INTERACTION)
        o.ibl_B, o.ibl_R1, o.ibl_R2 = r32(ff.IBL_SCALE), r32(R1), r32(R2)
        o.use_central = int(ff.CF_USE_CENTRAL_FORCE)
        if o.use_central:
            w = np.asarray(system.chrom_strength, dtype=np.float64)
            if as_float32_inputs:
                w = w.astype(np.float32).astype(np.float64)
            self._keep["w"] = np.ascontiguousarray(w)
            o.cf_w = self._keep["w"].ctypes.data
        o.cf_G, o.cf_R1 = r32(ff.CF_STRENGTH), r32(R1)
        c = system.centre
        for k in range(3):
            o.centre[k] = r32(c[k])
        o.use_chb = int(getattr(ff, "CHB_USE_CHROMOSOMAL_BLOCKS", False))
        if o.use_chb:
            self._keep["chrom"] = np.ascontiguousarray(system.chrom_of, dtype=np.int32)
            o.chrom_of = self._keep["chrom"].ctypes.data
            o.chb_kc, o.chb_de = r32(ff.CHB_KC), r32(ff.CHB_DE)
        # alternative functional forms (the *_FORCE_TYPE keys)
        o.ev_form = FORM_NAMES["EV_FORCE_TYPE"].index(getattr(ff, "EV_FORCE_TYPE", "powerlaw"))
        o.cob_form = FORM_NAMES["COB_FORCE_TYPE"].index(getattr(ff, "COB_FORCE_TYPE", "gaussian"))
        o.scb_form = FORM_NAMES["SCB_FORCE_TYPE"].index(getattr(ff, "SCB_FORCE_TYPE", "gaussian"))
        o.chb_form = FORM_NAMES["CHB_FORCE_TYPE"].index(getattr(ff, "CHB_FORCE_TYPE", "polynomial"))
        o.lam_form = FORM_NAMES["BLAMINA_FORCE_TYPE"].index(getattr(ff, "BLAMINA_FORCE_TYPE", "sin"))
        o.cf_form = FORM_NAMES["CENTRAL_FORCE_TYPE"].index(getattr(ff, "CENTRAL_FORCE_TYPE", "harmonic"))
        o.loop_form = FORM_NAMES["LE_LOOP_FORCE_TYPE"].index(getattr(ff, "LE_LOOP_FORCE_TYPE", "harmonic"))
        o.has_cob, o.has_scb = int(ff.COB_USE_COMPARTMENT_BLOCKS), int(ff.SCB_USE_SUBCOMPARTMENT_BLOCKS)
        for a in (1, 2):  # COB: Ea on {1,2}x{1,2}, Eb on {-1,-2}x{-1,-2} (model.py:246-253)
            for b in (1, 2):
                o.tab_cob[(a + 2) * 5 + b + 2] = r32(ff.COB_EA) if o.has_cob else 0.0
                o.tab_cob[(2 - a) * 5 + 2 - b] = r32(ff.COB_EB) if o.has_cob else 0.0
        if o.has_scb:      # SCB: diagonal only (model.py:322-333)
            o.tab_scb[4 * 5 + 4], o.tab_scb[3 * 5 + 3] = r32(ff.SCB_EA1), r32(ff.SCB_EA2)
            o.tab_scb[1 * 5 + 1], o.tab_scb[0 * 5 + 0] = r32(ff.SCB_EB1), r32(ff.SCB_EB2)
        self.o = o
        self.f32 = as_float32_inputs

    def _pos(self, positions=None) -> np.ndarray:
        p = self.s.positions if positions is None else positions
        p = np.asarray(p)
        if self.f32:
            p = p.astype(np.float32)
        return np.ascontiguousarray(p, dtype=np.float64).reshape(self.s.n_beads, 3)

    def eval(self, positions=None):
        """Returns (energy_terms[8], forces [N,3]) in fp64."""
        x = self._pos(positions)
        F = np.zeros_like(x)
        et = np.zeros(N_TERMS)
        rc = lib().orc_eval(C.byref(self.o), x.ctypes.data, F.ctypes.data, et.ctypes.data)
        if rc != 0:
            raise MemoryError("oracle allocation failed")
        return et, F

    def energy(self, positions=None) -> float:
        x = self._pos(positions)
        et = np.zeros(N_TERMS)
        lib().orc_eval(C.byref(self.o), x.ctypes.data, None, et.ctypes.data)
        return float(et.sum())

    def minimize(self, tolerance: float = 10.0, max_iters: int = 0, positions=None):
        """fp64 liblbfgs restatement.  Returns (positions [N,3], OrcMinStats)."""
        x = self._pos(positions).copy()
        st = OrcMinStats()
        rc = lib().orc_minimize(C.byref(self.o), x.ctypes.data, float(tolerance), int(max_iters), C.byref(st))
        if rc != 0:
            raise MemoryError("oracle allocation failed")
        return x, st


    # ---- the tuned CPU baseline (mmx_cpu_fast.c): fp32, SIMD, OpenMP; default functional forms, cutoff only ----------
    def fast_eval(self, positions=None):
        """(energy_terms, forces [N,3], lane-pairs swept) of the tuned fp32 CPU evaluation; raises NotImplementedError for
        systems it does not cover (non-default forms, NoCutoff, chromosomal blocks)."""
        x = self._pos(positions)
        F = np.zeros_like(x)
        et = np.zeros(N_TERMS)
        swept = C.c_double(0.0)
        rc = lib().orc_fast_eval(C.byref(self.o), x.ctypes.data, F.ctypes.data, et.ctypes.data, C.byref(swept))
        if rc == -2:
            raise NotImplementedError("the tuned CPU baseline covers the default functional forms with a cutoff only")
        if rc != 0:
            raise MemoryError("allocation failed")
        return et, F, swept.value

    def fast_minimize(self, tolerance: float = 10.0, max_iters: int = 0, positions=None):
        """The L-BFGS of ``minimize`` on the tuned evaluation.  Returns (positions, OrcMinStats, lane-pairs swept)."""
        x = self._pos(positions).copy()
        st = OrcMinStats()
        swept = C.c_double(0.0)
        rc = lib().orc_fast_minimize(C.byref(self.o), x.ctypes.data, float(tolerance), int(max_iters), C.byref(st), C.byref(swept))
        if rc == -2:
            raise NotImplementedError("the tuned CPU baseline covers the default functional forms with a cutoff only")
        if rc != 0:
            raise MemoryError("allocation failed")
        return x, st, swept.value

    def md_step(self, x, v, n_steps, kind="langevin", dt=0.001, temperature=310.0, friction=0.5,
                mass=16427.889, seed=0, step0=0, amd_alpha=100.0, amd_e=1000.0):
        """Advances copies of (x, v) by n_steps of the named integrator; returns (x, v, OrcMdStats).
        kind "amd": mm.amd.AMDIntegrator(dt, amd_alpha, amd_e) (model.py:794-800; config.py:255-256 defaults)."""
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(self.s.n_beads, 3)).copy()
        v = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(self.s.n_beads, 3)).copy()
        st = OrcMdStats()
        if kind == "amd":
            rc = lib().orc_md_step_amd(C.byref(self.o), float(dt), float(mass), float(amd_alpha), float(amd_e),
                                       int(step0), int(n_steps), x.ctypes.data, v.ctypes.data, C.byref(st))
            if rc != 0:
                raise MemoryError("oracle allocation failed")
            return x, v, st
        rc = lib().orc_md_step(C.byref(self.o), MD_KINDS[kind], float(dt), float(temperature), float(friction),
                               float(mass), int(seed), int(step0), int(n_steps), x.ctypes.data, v.ctypes.data,
                               C.byref(st))
        if rc != 0:
            raise MemoryError("oracle allocation failed")
        return x, v, st


def philox4x32_10(counter, key) -> np.ndarray:
    c = np.ascontiguousarray(counter, dtype=np.uint32)
    k = np.ascontiguousarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def normal3(bead: int, step: int, stream: int, seed: int) -> np.ndarray:
    out = np.zeros(3)
    lib().orc_normal3(bead, step, stream, seed, out.ctypes.data)
    return out


def md_velocities(n: int, temperature: float, mass: float, seed: int) -> np.ndarray:
    v = np.zeros((n, 3))
    lib().orc_md_velocities(n, float(temperature), float(mass), int(seed), v.ctypes.data)
    return v


def hilbert_points_c(n_points: int, p: int = 8, n: int = 3) -> np.ndarray:
    out = np.zeros((n_points, n), dtype=np.int32)
    lib().orc_hilbert_points(n_points, p, n, out.ctypes.data)
    return out


def backbone_flags_c(n_beads: int, chr_ends) -> np.ndarray:
    ce = np.ascontiguousarray(chr_ends, dtype=np.int32)
    out = np.zeros(n_beads, dtype=np.uint8)
    lib().orc_backbone_flags(n_beads, ce.ctypes.data, len(ce), out.ctypes.data)
    return out


def max_threads() -> int:
    return int(lib().orc_max_threads())


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def cpu_share() -> int:
    """Cores this process may actually use: the scheduler affinity mask, cut by the cgroup CPU quota when there is one (a
    GPU box hands a job 16 of its 128 hardware threads: 128 OpenMP threads on them only get in each other's way)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, math.ceil(q / per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)
