"""N > 1 path on CPU: two gloo ranks exercise the rendezvous helpers of multimm_amd/parallel.py and check
the decomposition identities the multi-GPU run relies on (slices partition the beads; per-rank energy
shares and force slices of the oracle recombine to the single-domain result)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from multimm_amd import synthetic_system
        from multimm_amd.parallel import broadcast_bytes, owned_share_of_energy, reduce_job_stats, slice_of
        from oracle import oracle_np
        from oracle.oracle import Oracle

        uid = bytes(range(128)) if rank == 0 else None
        got = broadcast_bytes(uid, 128)
        assert got == bytes(range(128))

        s = synthetic_system("gw_200k", n_beads=620, jitter=0.02, seed=0)   # 10 segments of 62 beads: two equal slices
        lo, hi = slice_of(s.n_beads, rank, world)
        et, F = Oracle(s, as_float32_inputs=False).eval()
        # forces: every rank contributes its slice, the gathered result is the full force field
        import torch
        mine = torch.from_numpy(np.ascontiguousarray(F[lo:hi]))
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        assert np.array_equal(torch.cat(parts).numpy(), F)
        # pair-energy shares: 1/2 * sum over owned beads of the full-shell sums add up to the total
        x = s.positions
        d = x[:, None, :] - x[None, :, :]
        r = np.sqrt((d * d).sum(-1))
        np.fill_diagonal(r, np.inf)
        e_ev = 100.0 * (0.1 / (r + 0.05)) ** 6
        e_ev[r >= 0.6] = 0.0
        share = torch.tensor([owned_share_of_energy(e_ev.sum(1), lo, hi)], dtype=torch.float64)
        dist.all_reduce(share)
        assert abs(share.item() - et[0]) <= 1e-9 * abs(et[0])
        tmax, iters = reduce_job_stats(1.0 + rank, 10, "ensemble")
        assert tmax == 2.0 and iters == 10 * world
        tmax, iters = reduce_job_stats(1.0 + rank, 10, "dd")
        assert iters == 10
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_decomposition():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_slices_partition_the_beads():
    from multimm_amd.parallel import slice_of
    for n, w in ((10, 3), (200000, 8), (1000000, 8), (7, 7), (9, 8)):
        cover = []
        for r in range(w):
            lo, hi = slice_of(n, r, w)
            assert 0 <= lo <= hi <= n
            cover += list(range(lo, hi)) if n < 100 else [lo, hi]
        if n < 100:
            assert cover == list(range(n))
    with pytest.raises(ValueError):
        slice_of(10, 3, 3)
