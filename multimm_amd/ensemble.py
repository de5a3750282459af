"""Ensemble generation (BASELINE config 4): the replica loop of ``run.py:471-485`` on ``PLATFORM = MI355X``.

The reference runs ``N_ENSEMBLE`` replicas one after the other in one process, each with ``SHUFFLING_SEED = i`` and
``OUT_PATH = <name>/run_<i, zero padded>``, and compresses every run directory afterwards (``archive_run``,
run.py:423-445).  The replicas are independent, so with several GPUs replica i goes to rank i mod world (one
process per GPU, no data-path collective: what ``bench.py --gpus N`` times).  Small systems leave most of a GPU
idle (one replica of 5 000 beads is bound by launch latency), so a rank may also keep ``concurrent`` replicas in
flight on its GPU, each with its own engine handle, stream and host thread: measured 2x aggregate iterations/s with
two and 2.9x with three replicas of 5 000 beads on one MI355X, +39 % with three of 200 000 beads; beyond three the
streams share hardware queues and the per-replica rate halves (``scripts/concurrent_replicas.py``).
"""
from __future__ import annotations

import copy
import os
import shutil
import tarfile
from typing import Optional

from .config import SimulationConfig, load_config
from .model import MultiMM


def archive_run(run_path: str) -> str:
    """<run_path>.tar.gz of the run directory, then the directory is removed (run.py:423-445)."""
    tar_path = run_path + ".tar.gz"
    with tarfile.open(tar_path, "w:gz") as tar:
        tar.add(run_path, arcname=os.path.basename(run_path))
    if not (os.path.exists(tar_path) and os.path.getsize(tar_path) > 0):
        raise RuntimeError(f"Archive creation failed ({tar_path}). Original directory was NOT deleted.")
    shutil.rmtree(run_path)
    return tar_path


def run_ensemble(args: SimulationConfig | str | dict, n_ensemble: Optional[int] = None, rank: int = 0, world: int = 1,
                 archive: bool = True, device: Optional[int] = None, concurrent: int = 1, observer=None,
                 **model_inputs) -> list:
    """Runs replicas ``i = rank, rank + world, ...`` of ``n_ensemble`` (default ``args.N_ENSEMBLE``); returns
    ``[(i, run_path_or_archive, stats)]`` of the replicas this rank ran, in replica order.  ``model_inputs`` are
    passed to ``MultiMM`` (``ms, ns, ds, chr_ends, Cs``) when the tensors are given instead of files.
    ``concurrent`` > 1 keeps that many replicas in flight on this rank's GPU (threads; the library calls release
    the GIL); every replica's result is the one it has when run alone (replicas share nothing).
    ``observer(i, stage, model)``: see ``MultiMM.run``."""
    base = args if isinstance(args, SimulationConfig) else load_config(args)
    n = int(n_ensemble if n_ensemble is not None else (base.N_ENSEMBLE or 1))
    name = base.OUT_PATH
    width = len(str(max(n - 1, 0)))
    def one(i: int):
        cfg = copy.deepcopy(base)
        cfg.SHUFFLING_SEED = i
        cfg.DEVICE = device if device is not None else (base.DEVICE if world == 1 else rank)  # one GPU per rank
        run_path = os.path.join(name, f"run_{i:0{width}d}")
        cfg.OUT_PATH = run_path
        os.makedirs(run_path, exist_ok=True)
        md = MultiMM(cfg, **model_inputs)
        try:
            stats = md.run(observer=None if observer is None else (lambda stage, m: observer(i, stage, m)))
        finally:
            if md.engine is not None:
                md.engine.close()
        return (i, archive_run(run_path) if archive else run_path, stats)

    mine = list(range(rank, n, max(world, 1)))
    if concurrent <= 1 or len(mine) <= 1:
        return [one(i) for i in mine]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=int(concurrent)) as pool:
        return list(pool.map(one, mine))


def run(args: SimulationConfig | str | dict, rank: int = 0, world: int = 1, concurrent: int = 1, **model_inputs):
    """The dispatch of ``run.py:469-491``: ``GENERATE_ENSEMBLE`` -> the replica loop (``run_ensemble``), else one
    ``MultiMM(args).run()``; returns what the branch taken returns."""
    cfg = args if isinstance(args, SimulationConfig) else load_config(args)
    if cfg.GENERATE_ENSEMBLE:
        if not cfg.N_ENSEMBLE:
            raise ValueError("GENERATE_ENSEMBLE needs N_ENSEMBLE")
        return run_ensemble(cfg, rank=rank, world=world, concurrent=concurrent, **model_inputs)
    md = MultiMM(cfg, **model_inputs)
    try:
        return md.run()
    finally:
        if md.engine is not None:
            md.engine.close()
