"""multimm_amd -- MI355X-native force-field + L-BFGS minimizer behind MultiMM's simulation entry point.

Only the hot path of SFGLab/MultiMM is rebuilt here: what
``MultiMM.min_energy()`` (reference ``src/multimm/model.py:859-897``) obtains from OpenMM for the
force terms ``MultiMM.add_forcefield()`` installs (``model.py:812-857``).  Host code is Python
(numpy + ctypes); all arithmetic on the path runs in hand-written HIP kernels inside
``libmmx.so`` (``multimm_amd/csrc``), reached through the C ABI of ``include/mmx.h``.
There is no CPU fallback: without the built library and a gfx950 GPU the engine raises.
"""
from .system import ChromatinSystem, ForceFieldParams, synthetic_system, backbone_flags  # noqa: F401
from .hilbert import hilbert_points  # noqa: F401

__all__ = ["ChromatinSystem", "ForceFieldParams", "synthetic_system", "backbone_flags", "hilbert_points"]
__version__ = "0.1.0"
