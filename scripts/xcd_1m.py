"""gw_1m: full-shell kernel with and without the XCD-slab mapping, half-shell kernel with items taken in descending / ascending order.
usage: xcd_1m.py"""
import sys
sys.path.insert(0, '.')
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED
eng = engine_for(synthetic_system("gw_1m"))
done = 0
for upto in (0, 300):
    if upto > done:
        eng.set_option("nb_variant", 0); eng.minimize(tolerance=0.0, max_iters=upto - done); done = upto
    for v, nm in ((8192, "full shell"), (8192 + 2048, "full shell, XCD slabs"), (4096, "half shell"), (4096 + (64 << 16), "half shell, ascending items")):
        eng.set_option("nb_variant", v)
        print(f"gw_1m after {done}: {nm}: {eng.time_kernel(K_NONBONDED, 10)[0]:.1f} us", flush=True)
