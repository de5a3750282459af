"""A force kernel that cannot do its work must end the call with an error, not with partial forces.

The half-shell pair kernel (k_nb_n3) has two such exits: a wave of its work-unit pipeline gives up waiting (a protocol
bug, bounded spin-wait) and a cell build that needs more work items than the list holds.  Neither can be provoked by
data, so the option "inject_fault" provokes them: bit 0 makes every wait time out at once, bit 1 shrinks the item list
to one entry.  Partial force sums are finite -- without the error word the minimizer would carry on with a wrong
gradient and return MMX_OK (round 2 did)."""
import numpy as np
import pytest

from multimm_amd import synthetic_system
from multimm_amd.engine import MMXError, engine_for

pytestmark = pytest.mark.gpu
MMX_ERR_STATE = -5


@pytest.mark.parametrize("fault", [1, 2])
def test_pair_kernel_failure_surfaces_from_every_entry_point(fault):
    s = synthetic_system("gw_200k", n_beads=20000, jitter=0.02, seed=3)
    with engine_for(s) as eng:
        eng.set_option("nb_variant", 4096)          # the half-shell kernel, whatever the size
        et0, f0 = eng.compute()
        assert np.isfinite(et0).all() and eng.get_option("n3_launches") > 0
        eng.set_option("inject_fault", fault)
        with pytest.raises(MMXError) as exc:
            eng.compute()
        assert exc.value.code == MMX_ERR_STATE and "k_nb_n3" in str(exc.value)
        with pytest.raises(MMXError) as exc:
            eng.minimize(tolerance=0.0, max_iters=5)
        assert exc.value.code == MMX_ERR_STATE
        eng.md_configure("verlet", dt_ps=0.001)
        with pytest.raises(MMXError) as exc:
            eng.md_step(3)
        assert exc.value.code == MMX_ERR_STATE
        # the handle survives: without the fault the same calls work again and give the same answer
        eng.set_option("inject_fault", 0)
        eng.set_positions(s.positions)              # (the void MD steps moved the beads with partial forces)
        et1, f1 = eng.compute()
        assert np.allclose(et1, et0, rtol=1e-6, atol=1e-3)
        assert np.abs(f1 - f0).max() <= 1e-5 * np.abs(f0).max()
        st = eng.minimize(tolerance=0.0, max_iters=5)
        assert st.iterations == 5 and st.status == 1
