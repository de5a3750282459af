"""Idle tail of the half-shell kernel: per-workgroup and per-wave exit times from a -DMMX_N3_TIMING build of libmmx.so (loaded through
MMX_LIB), units and later passes per launch.  usage: MMX_LIB=<timing build> n3_tail.py [nb_variant bits 24-30 = tail configuration]"""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for, K_NONBONDED, load_library
lib = load_library()
eng = engine_for(synthetic_system("gw_200k"))
done = 0
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # nb_variant bits 24-30 (64: no tail shares)
for upto in (0, 10, 30, 60, 100, 400, 2000):
    if upto > done:
        eng.minimize(tolerance=0.0, max_iters=upto - done); done = upto
    eng.set_option("nb_variant", 4096 + (cfg << 24))
    t = eng.time_kernel(K_NONBONDED, 5)[0]
    buf = np.zeros(512 * 20, np.uint64)
    lib.mmx_debug_n3_times(C.c_void_p(buf.ctypes.data))
    b = buf.reshape(512, 20)[:256].astype(np.int64)
    t0 = b[:, 0].min()
    start = (b[:, 0] - t0) / 100.0   # us (100 MHz)
    wend = (b[:, 1:17] - t0) / 100.0
    bend = (b[:, 17] - t0) / 100.0
    print(f"units {int(buf.reshape(512, 20)[:256, 19].sum())}, of them later passes of an item {int(buf.reshape(512, 20)[:256, 18].sum())}")
    print(f"after {done} it: kernel {t:.1f} us; block start {start.min():.1f}..{start.max():.1f}; wave exit: min {wend.min():.1f} mean {wend.mean():.1f} max {wend.max():.1f}; "
          f"block end: min {bend.min():.1f} p10 {np.percentile(bend,10):.1f} median {np.median(bend):.1f} p90 {np.percentile(bend,90):.1f} max {bend.max():.1f}; "
          f"mean idle of waves before the last one ends: {(wend.max() - wend).mean():.1f} us", flush=True)
    eng.set_option("nb_variant", 0)
