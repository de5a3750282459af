"""BASELINE config 4 at its own size: 8 independent genome-wide replicas (200 000 beads, GW preset force set, seeds 0..7)
through the replica loop of ``run.py:473-485`` (``multimm_amd.ensemble.run_ensemble``: SHUFFLING_SEED = i, OUT_PATH =
<name>/run_<i>), every one of them checked against the fp64 oracle -- per-term energies and forces at the start and on the
state after 20 L-BFGS iterations -- at the tolerances of tests/test_gpu_parity.py.  On an 8-GPU node replica i runs on rank
i (``rank=i, world=8``: no data-path collective); here the one GPU takes them one after the other, which is what the
reference itself does.  The replicas are the systems ``bench.py --gpus N`` times (``synthetic_system("gw_200k", seed=rank)``)."""
import os

import numpy as np
import pytest

from multimm_amd import synthetic_system
from multimm_amd.engine import TERM_NAMES
from multimm_amd.ensemble import run_ensemble

pytestmark = pytest.mark.gpu

E_RTOL, E_ATOL = 2e-6, 1e-3
F_RTOL, F_ATOL = 4e-6, 2e-3
E_RTOL_AT_CUTOFF, F_RTOL_AT_CUTOFF = 3e-5, 1e-5       # the lattice start: pairs exactly at r_c (see test_gpu_parity.py)

GW_PRESET = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, IBL_USE_B_LAMINA_INTERACTION=True)


def _compare(model, positions, e_rtol, f_rtol, tag):
    from oracle.oracle import Oracle
    et_ref, F_ref = Oracle(model.system).eval(positions)
    et, F = model.engine.compute()
    scale_e, fmax = np.abs(et_ref).sum(), np.abs(F_ref).max()
    for t in range(len(TERM_NAMES)):
        assert abs(et[t] - et_ref[t]) <= e_rtol * scale_e + E_ATOL, (tag, TERM_NAMES[t], et[t], et_ref[t])
    ferr = np.abs(F.astype(np.float64) - F_ref).max()
    print(f"ACC {tag}: E_total {et.sum():.6e} (ref {et_ref.sum():.6e}), max force err / max|F| = {ferr / fmax:.2e}")
    assert ferr <= f_rtol * fmax + F_ATOL, (tag, ferr, fmax)
    return float(et_ref.sum())


def test_config4_eight_genome_wide_replicas_against_the_oracle(tmp_path):
    cfg = dict(PLATFORM="MI355X", N_BEADS=200_000, OUT_PATH=str(tmp_path / "ens"), N_ENSEMBLE=8, NB_CUTOFF=0.6,
               MIN_MAX_ITERATIONS=20, **GW_PRESET)
    seen = {}

    def observer(i, stage, m):
        if stage == "forcefield":
            twin = synthetic_system("gw_200k", seed=i)      # what bench.py's ensemble leg hands to rank i
            assert np.array_equal(m.system.labels, twin.labels) and np.array_equal(m.system.loop_m, twin.loop_m)
            assert np.array_equal(m.system.loop_n, twin.loop_n) and np.array_equal(m.system.loop_r0, twin.loop_r0)
            assert np.array_equal(m.system.chr_ends, twin.chr_ends) and m.system.ff == twin.ff
            assert np.allclose(m.system.positions, twin.positions, atol=1e-12)
            e0 = _compare(m, None, E_RTOL_AT_CUTOFF, F_RTOL_AT_CUTOFF, f"replica {i} start")
            seen[i] = [e0]
        else:
            assert m.stats.iterations == 20
            e1 = _compare(m, m.state_positions, E_RTOL, F_RTOL, f"replica {i} after 20 iterations")
            assert abs(e1 - m.stats.e_final) <= 2e-6 * abs(e1) + 1e-2
            seen[i].append(e1)

    res = run_ensemble(cfg, archive=False, observer=observer)
    assert [i for i, _, _ in res] == list(range(8)) and sorted(seen) == list(range(8))
    for i, path, st in res:
        assert path.endswith(f"run_{i}") and os.path.exists(os.path.join(path, "model", "MultiMM_minimized.cif"))
        assert st.e_final < st.e_initial and seen[i][1] < seen[i][0]
    # different seeds are different systems (labels and loops differ): no two replicas share an energy
    assert len({round(e[0], 1) for e in seen.values()}) == 8 and len({round(e[1], 1) for e in seen.values()}) == 8
    # a rank of an 8-GPU job takes exactly its own replica
    mine = run_ensemble(dict(cfg, OUT_PATH=str(tmp_path / "rank5"), N_BEADS=20_000), rank=5, world=8, archive=False, device=0)
    assert [i for i, _, _ in mine] == [5]
