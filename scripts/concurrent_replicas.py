"""Aggregate minimizer throughput of K replicas sharing ONE GPU (one handle + stream + host thread each): what the
ensemble loop (run.py:471-485) can gain for small systems, whose single replica leaves most CUs idle.
usage: concurrent_replicas.py [workload=region_5k] [n_beads=0] [iters=2000] [K,K,...=1,2,4,8]"""
import sys, time, threading
sys.path.insert(0, ".")
from multimm_amd import synthetic_system
from multimm_amd.engine import engine_for

name = sys.argv[1] if len(sys.argv) > 1 else "region_5k"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
ks = [int(k) for k in (sys.argv[4] if len(sys.argv) > 4 else "1,2,4,8").split(",")]
for K in ks:
    systems = [synthetic_system(name, seed=i, n_beads=nb or None) for i in range(K)]
    engines = [engine_for(s) for s in systems]
    for e in engines:
        e.minimize(tolerance=0.0, max_iters=20)   # warm-up
    done = [0] * K
    start = threading.Barrier(K + 1)

    def work(i):
        start.wait()
        done[i] = engines[i].minimize(tolerance=0.0, max_iters=iters).iterations

    th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for e in engines:
        e.close()
    print(f"{name} n={systems[0].n_beads} K={K}: {sum(done)} iterations in {dt:.3f} s = {sum(done) / dt:.0f} iterations/s aggregate "
          f"({sum(done) / dt / K:.0f} per replica)", flush=True)
