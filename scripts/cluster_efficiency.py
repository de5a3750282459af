"""Offline (CPU, numpy): how many of the lanes the pair kernels sweep lie inside the cutoff, for alternative per-bead culls and
cluster shapes, on positions dumped by scripts/dump_states.py.  Cells of edge r_c, in-cell order along a Hilbert curve of 16^3
sub-cells, clusters of 8 consecutive beads (what k_cell_order builds); a sample of i-clusters against every bead of their 27 cells.
usage: cluster_efficiency.py gpurun_out/pos_400.npy [sample=3000]"""
import sys
import numpy as np
rc = 0.6
x = np.load(sys.argv[1]).astype(np.float64)
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
n = len(x)
lo = x.min(0); h = rc * 1.001
cell3 = np.floor((x - lo) / h).astype(np.int64)
dims = cell3.max(0) + 1
cid = (cell3[:, 2] * dims[1] + cell3[:, 1]) * dims[0] + cell3[:, 0]


def hilbert12(q):  # Skilling, 3 axes x 4 bits, vectorised (mmx_cells.hpp: hilbert12)
    X = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
    Q = 8
    while Q > 1:
        P = Q - 1
        for i in range(3):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = np.where(m, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t; X[i] ^= t
        Q >>= 1
    X[1] ^= X[0]; X[2] ^= X[1]
    t = np.zeros_like(X[0]); Q = 8
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t); Q >>= 1
    X = [v ^ t for v in X]
    out = np.zeros_like(X[0])
    for b in range(4):
        for a in range(3):
            out |= ((X[a] >> b) & 1) << (3 * b + (2 - a))
    return out


sub = np.clip(((x - lo) / h - cell3) * 16, 0, 15).astype(np.int64)
key = hilbert12(sub)
order = np.lexsort((np.arange(n), key, cid))
cs = cid[order]
starts = np.searchsorted(cs, np.arange(dims.prod() + 1))
rng = np.random.default_rng(0)


def clusters_of_cell(c, mode):
    idx = order[starts[c]:starts[c + 1]]
    if mode == "kd" and len(idx) > 8:  # recursive median split along the longest axis down to <= 8 beads
        out = []
        stack = [idx]
        while stack:
            g = stack.pop()
            if len(g) <= 8:
                out.append(g); continue
            p = x[g]; ax = np.argmax(p.max(0) - p.min(0))
            g = g[np.argsort(p[:, ax], kind="stable")]
            half = ((len(g) + 15) // 16) * 8 if len(g) > 8 else len(g)
            stack.append(g[:half]); stack.append(g[half:])
        return out
    return [idx[k:k + 8] for k in range(0, len(idx), 8)]


def neighbours(c):
    cx, cy, cz = c % dims[0], (c // dims[0]) % dims[1], c // (dims[0] * dims[1])
    out = []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                X, Y, Z = cx + dx, cy + dy, cz + dz
                if 0 <= X < dims[0] and 0 <= Y < dims[1] and 0 <= Z < dims[2]:
                    cc = (Z * dims[1] + Y) * dims[0] + X
                    out.append(order[starts[cc]:starts[cc + 1]])
    return np.concatenate(out)


occupied = np.unique(cid)
for mode in ("hilbert", "kd"):
    tot = dict(pairs=0, box=0, boxsph=0, quad=0, edge=0.0, ncl=0)
    for c in rng.choice(occupied, min(len(occupied), nsamp // 10), replace=False):
        nb = x[neighbours(c)]
        cl = clusters_of_cell(c, mode)
        for g in [cl[k] for k in rng.choice(len(cl), min(len(cl), 10), replace=False)]:
            p = x[g]
            blo, bhi = p.min(0), p.max(0)
            d = np.maximum(np.maximum(blo - nb, nb - bhi), 0.0)
            inbox = (d * d).sum(1) < rc * rc
            cen = p.mean(0); rho = np.sqrt(((p - cen) ** 2).sum(1)).max()
            insph = ((nb - cen) ** 2).sum(1) < (rc + rho) ** 2
            r2 = ((p[:, None, :] - nb[None, inbox, :]) ** 2).sum(-1)
            tot["pairs"] += int((r2 < rc * rc).sum())
            tot["box"] += 8 * int(inbox.sum())
            tot["boxsph"] += 8 * int((inbox & insph).sum())
            # two quads with boxes of their own: a bead is swept 4 times per quad whose box it is near
            for quad in (p[:4], p[4:]):
                if len(quad):
                    dq = np.maximum(np.maximum(quad.min(0) - nb, nb - quad.max(0)), 0.0)
                    tot["quad"] += 4 * int(((dq * dq).sum(1) < rc * rc).sum())
            tot["edge"] += float((bhi - blo).mean()); tot["ncl"] += 1
    print(f"{sys.argv[1]} {mode:8s}: mean box edge {tot['edge'] / tot['ncl']:.3f} nm; lanes inside the cutoff: box cull {tot['pairs'] / tot['box']:.3f}, "
          f"box + bounding sphere {tot['pairs'] / tot['boxsph']:.3f}, two quads of four {tot['pairs'] / tot['quad']:.3f}")
