"""Same-box A/B of two builds of the library on frozen ranks of a decomposed run: force evaluation as launched + list / halo kernels.
usage: dd_ab_libs.py <libA> <libB> [workload=gw_1m] [world=8] [relax=150]   (runs itself once per library)"""
import sys, os, subprocess, threading
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, '.')
    from multimm_amd import synthetic_system
    from multimm_amd.engine import Engine, engine_for, K_FORCES, K_DD_LISTS, K_CELL_BUILD
    name, world, relax = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    s = synthetic_system(name)
    engines = [engine_for(s, rank=r, world=world) for r in range(world)]
    Engine.comm_init_local(engines)
    def work(e):
        e.minimize(tolerance=0.0, max_iters=relax); e.compute()
    th = [threading.Thread(target=work, args=(e,)) for e in engines]
    [t.start() for t in th]; [t.join() for t in th]
    out = []
    for e in engines:
        e.set_option("dd_freeze", 1)
        f, l, b = e.time_kernel(K_FORCES, 20)[0], e.time_kernel(K_DD_LISTS, 20)[0], e.time_kernel(K_CELL_BUILD, 20)[0]
        out.append((f + l, b))
    print(" ".join(f"{c:6.1f}/{b:4.1f}" for c, b in out), "| slowest critical path", f"{max(c for c, _ in out):.1f}")
    for e in engines: e.close()
    sys.exit(0)
libs = sys.argv[1:3]
rest = sys.argv[3:] + ["gw_1m", "8", "150"][len(sys.argv) - 3:]
for rep in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, __file__, "--one"] + rest, env=dict(os.environ, MMX_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print(os.path.basename(lib), "critical path / cell-build slot per rank (us):", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
