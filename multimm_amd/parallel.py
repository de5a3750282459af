"""Host-side helpers for multi-GPU runs (one process per GPU, torch.distributed for rendezvous only).

Two modes:
  * ensemble -- BASELINE config 4: independent replicas with seeds 0..N-1, no data-path collective
    (the reference's ensemble loop run.py:471-485 is sequential and embarrassingly parallel);
  * dd       -- BASELINE config 5: one system, bead slices owned by the ranks; the library itself issues
    the RCCL collectives on its stream (include/mmx.h, "multi-GPU").  torch.distributed only carries
    the 128-byte ncclUniqueId to the ranks and the final timing reduction.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


SEGMENT = 62   # beads per ownership segment (kSeg, csrc/mmx_common.hpp: one backbone tile)


def slice_of(n_beads: int, rank: int, world: int) -> Tuple[int, int]:
    """Bead range [lo, hi) `rank` owns when a decomposed handle is created -- the rule of mmx_create_dd: one rank alone owns
    everything; otherwise equal slices of 62 * ceil(ceil(N / 62) / world) beads (whole 62-bead segments), the last one(s)
    clipped to N.  (mmx_minimize re-assigns segments afterwards: ask Engine.owned_beads().)"""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("rank/world out of range")
    if world == 1:
        return 0, n_beads
    nseg = (n_beads + SEGMENT - 1) // SEGMENT
    s = SEGMENT * ((nseg + world - 1) // world)
    lo = min(n_beads, rank * s)
    return lo, min(n_beads, lo + s)


def broadcast_bytes(payload: bytes | None, nbytes: int, src: int = 0, device=None) -> bytes:
    """Broadcast a fixed-size byte string from `src` to every rank of the default process group."""
    import torch
    import torch.distributed as dist
    t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        t.copy_(torch.tensor(list(payload), dtype=torch.uint8))
    dist.broadcast(t, src=src)
    return bytes(t.cpu().tolist())


def reduce_job_stats(seconds: float, iterations: int, mode: str, device=None) -> Tuple[float, float]:
    """(max wall time over ranks, job iterations): replicas add up, a decomposed system counts once."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    it = torch.tensor([float(iterations)], dtype=torch.float64, device=device)
    dist.all_reduce(it, op=dist.ReduceOp.SUM if mode == "ensemble" else dist.ReduceOp.MAX)
    return float(t.item()), float(it.item())


def owned_share_of_energy(per_bead_pair_energy: np.ndarray, lo: int, hi: int) -> float:
    """Energy share of a rank for pair terms: half of the full-shell sums of its owned beads, so that
    the shares of all ranks add up to the total with no special casing of cross-rank pairs."""
    return 0.5 * float(per_bead_pair_energy[lo:hi].sum())
