"""Offline comparison of ownership rules for the decomposed run: ghosts per rank under the engine's
need-map rule (coarse cells of edge e, maps grown by ceil(reach / e) cells) for
  (a) contiguous index slices (rounds 1-3), (b) segments of L consecutive beads assigned by recursive coordinate bisection of
  their centroids, on positions dumped by scripts/dump_states_1m.py.   usage: dd_ownership_offline.py pos.npy [world=8]"""
import sys
import numpy as np
from scipy import ndimage

pos = np.load(sys.argv[1]).astype(np.float64)
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = len(pos)
reach = 0.6


def rcb(cent, ids, ranks, out):
    """segments `ids` -> ranks [ranks[0], ranks[1]) by recursive bisection along the longest axis, sizes proportional"""
    nr = ranks[1] - ranks[0]
    if nr == 1:
        out[ids] = ranks[0]
        return
    c = cent[ids]
    ax = int(np.argmax(c.max(0) - c.min(0)))
    order = ids[np.argsort(c[:, ax], kind="stable")]
    nl = nr // 2
    cut = (len(order) * nl + nr // 2) // nr
    rcb(cent, order[:cut], (ranks[0], ranks[0] + nl), out)
    rcb(cent, order[cut:], (ranks[0] + nl, ranks[1]), out)


def ghosts(owner, gridn, label):
    lo = pos.min(0)
    ext = (pos.max(0) - lo).max()
    edge = max(0.5 * reach if gridn == 64 else reach / 4, ext / (gridn - 4))
    radius = max(1, int(np.ceil(reach / edge - 1e-4)))
    cell = np.floor((pos - (lo - 2 * edge)) / edge).astype(np.int64)
    cell = np.clip(cell, 0, gridn - 1)
    flat = (cell[:, 2] * gridn + cell[:, 1]) * gridn + cell[:, 0]
    st = np.ones((2 * radius + 1,) * 3, bool)
    res = []
    for r in range(world):
        occ = np.zeros(gridn ** 3, bool)
        occ[flat[owner == r]] = True
        need = ndimage.binary_dilation(occ.reshape(gridn, gridn, gridn), structure=st).reshape(-1)
        res.append(int((need[flat] & (owner != r)).sum()))
    res = np.array(res)
    own = np.bincount(owner, minlength=world)
    print(f"{label:44s} grid {gridn}^3 edge {edge:.3f} r {radius}: owned {own.min()}-{own.max()}  ghosts min/mean/max "
          f"{res.min()}/{res.mean():.0f}/{res.max()}  -> {res.mean() * 16 / 1e6:.2f} MB/rank/eval (exact lists)", flush=True)


slice_ = (n + world - 1) // world
for gridn in (64, 128):
    ghosts(np.arange(n) // slice_, gridn, "index slices")
    for L in (62, 124, 248, 496, 992, 1984):
        ns = (n + L - 1) // L
        seg = np.arange(n) // L
        cent = np.stack([np.bincount(seg, weights=pos[:, k], minlength=ns) for k in range(3)], 1) / np.bincount(seg, minlength=ns)[:, None]
        so = np.zeros(ns, np.int64)
        rcb(cent, np.arange(ns), (0, world), so)
        ghosts(so[seg], gridn, f"RCB of {L}-bead segments")
    # ideal: per-bead RCB (no segment constraint)
    so = np.zeros(n, np.int64)
    rcb(pos, np.arange(n), (0, world), so)
    ghosts(so, gridn, "RCB of single beads (bound)")
