"""The REAL RCCL path with more than one rank on a ONE-GPU box: every rank is a process of its own on device 0 and tells RCCL
it sits on a different host (NCCL_HOSTID), so RCCL's duplicate-GPU check does not apply and the ranks talk through its socket
transport over the loopback interface.  Functional check only (sockets, not xGMI): the library's ncclAllGather / grouped
ncclSend+ncclRecv / ncclAllReduce call sites, message capacities, halts -- against a single-domain engine.
usage: rccl_ranks_one_gpu.py [world=2] [n_beads=12000] [iters=30]      (environment: MMX_NB_VARIANT, MMX_DD_EVERY, MMX_DD_SKIN,
MMX_INJECT = engine options nb_variant, dd_rebuild_every, dd_skin, inject_fault).  Exit code 3: RCCL could not be initialised this
way on this machine (no loopback interface ...), 0: results equal the single-domain engine's."""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, n_beads, iters, q):
    os.environ.update({"NCCL_HOSTID": f"mmx-rank-{rank}", "NCCL_IB_DISABLE": "1", "NCCL_SOCKET_IFNAME": "lo",
                       "NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1", "NCCL_NET": "Socket",
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multimm_amd import synthetic_system
        from multimm_amd.engine import Engine, engine_for
        from multimm_amd.parallel import broadcast_bytes
        s = synthetic_system("gw_200k", n_beads=n_beads, jitter=0.02, seed=3)
        eng = engine_for(s, device=0, rank=rank, world=world)
        eng.set_option("nb_variant", float(os.environ.get("MMX_NB_VARIANT", "0")))
        for key, env in (("dd_rebuild_every", "MMX_DD_EVERY"), ("dd_skin", "MMX_DD_SKIN"), ("inject_fault", "MMX_INJECT"),
                         ("dd_reassign_first", "MMX_DD_REASSIGN_FIRST"), ("dd_reassign_max", "MMX_DD_REASSIGN_MAX")):
            if os.environ.get(env):
                eng.set_option(key, float(os.environ[env]))
        uid = broadcast_bytes(Engine.comm_unique_id() if rank == 0 else None, 128)
        try:
            eng.comm_init(uid)
        except Exception as e:  # noqa: BLE001 -- no loopback interface, no socket transport: an environment, not a result
            q.put((rank, "norccl", repr(e)))
            return
        et, f = eng.compute()
        ids = eng.owned_beads()
        st = eng.minimize(tolerance=0.0, max_iters=iters)
        x = eng.get_positions()
        stats = {k: eng.get_option(k) for k in ("dd_ghosts", "dd_exchanges", "dd_bytes_sent", "dd_halts", "dd_sync_rebuilds",
                                                "dd_reassignments", "dd_segments_moved")}
        # after a re-assignment of the segments the forces of the owned beads must still be the single-domain ones
        et2, f2 = eng.compute()
        ids2 = eng.owned_beads()
        q.put((rank, "ok", et, f, ids, len(ids), (st.iterations, st.status, st.e_initial, st.e_final), x, stats, et2, f2, ids2))
        eng.close()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", repr(e)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    import numpy as np
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n_beads = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, n_beads, iters, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=600) for _ in procs)
    [p.join(120) for p in procs]
    if any(r[1] == "norccl" for r in res):
        print("RCCL could not be initialised with several ranks on one GPU here:", [r for r in res if r[1] == "norccl"][:1]); sys.exit(3)
    bad = [r for r in res if r[1] != "ok"]
    if bad:
        print("FAILED:", bad); sys.exit(1)
    from multimm_amd import synthetic_system
    from multimm_amd.engine import engine_for
    s = synthetic_system("gw_200k", n_beads=n_beads, jitter=0.02, seed=3)
    with engine_for(s) as eng:
        eng.set_option("nb_variant", float(os.environ.get("MMX_NB_VARIANT", "0")))
        et0, F0 = eng.compute()
        st0 = eng.minimize(tolerance=0.0, max_iters=iters)
    F = np.zeros_like(F0)
    for r in res:
        F[r[4]] = r[3]
    print(f"RCCL, {world} ranks (processes) on one GPU, {n_beads} beads: energies equal on every rank: "
          f"{all(np.array_equal(r[2], res[0][2]) for r in res)}; max |dE| vs one domain {np.abs(res[0][2] - et0).max():.3g}; "
          f"max |dF| / max |F| {np.abs(F - F0).max() / np.abs(F0).max():.3g}; minimization {res[0][6]} vs {(st0.iterations, st0.status, st0.e_initial, st0.e_final)}; "
          f"positions equal on every rank: {all(np.array_equal(r[7], res[0][7]) for r in res)}; rank 0 halo: {res[0][8]}")
    with engine_for(s) as eng:      # forces at the END of the decomposed run, against one domain at those positions
        eng.set_positions(res[0][7])
        et1, F1 = eng.compute()
    F2 = np.zeros_like(F1)
    for r in res:
        F2[r[11]] = r[10]
    cover = sorted(np.concatenate([r[11] for r in res]).tolist()) == list(range(n_beads))
    print(f"after the run: ranks partition the beads: {cover}; segment re-assignments {res[0][8]['dd_reassignments']:.0f} "
          f"({res[0][8]['dd_segments_moved']:.0f} segments moved); max |dF| / max |F| at the end {np.abs(F2 - F1).max() / np.abs(F1).max():.3g}; "
          f"max |dE| {np.abs(res[0][9] - et1).max():.3g}")
    ok = (np.abs(F - F0).max() <= 1e-5 * np.abs(F0).max() and all(r[6] == res[0][6] for r in res)
          and all(np.array_equal(r[7], res[0][7]) for r in res) and cover and np.abs(F2 - F1).max() <= 1e-5 * np.abs(F1).max())
    sys.exit(0 if ok else 2)
